"""Input side of `varKoder query`: images that are already on the GPU -> the
float tensors the reference's fastai pipeline would feed its model, then a batched forward.

Reference: commands/query.py:283-324 (test_dl + get_preds, sigmoid >= threshold for multilabel),
item transform commands/train.py:236-245 (Resize(squish, BOX, BOX) to the timm model's fixed input
size), IntToFloatTensor (/255) and Normalize (0.5 / 0.5 for the default ViT,
xtra_scripts/push_to_hf.py:39-45).  The resize is PIL's 8-bit BOX resample, restated here as
coefficient tables (PIL's own precompute_coeffs + normalize_coeffs_8bpc arithmetic) consumed by
vk_preprocess_kernel.  fastai / timm / the pretrained weights are not available offline: the model
is whatever torch.nn.Module the caller supplies (vit_l32() builds the default architecture's shape
with random weights for throughput work).
"""
import ctypes as C

import numpy as np

from . import _capi

PRECISION_BITS = 32 - 8 - 2  # PIL Resample.c


def box_tables(in_size, out_size):
    """(bounds int32[out,2], coef int32[out,kmax]) of PIL's BOX filter for one axis."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 0.5 * filterscale
    kmax = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, kmax), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = np.zeros(xmax)
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            k[x] = 1.0 if -0.5 < t <= 0.5 else 0.0
        ww = k.sum()
        if ww != 0.0:
            k = k / ww
        bounds[xx] = (xmin, xmax)
        coef[xx, :xmax] = [int(0.5 + v * (1 << PRECISION_BITS)) for v in k]
    return bounds, coef


def preprocess(engine, images, out_size=224, mean=0.5, std=0.5):
    """uint8 device tensor [n, side, side] -> float32 device tensor [n, 3, out, out]."""
    import torch
    n, side, side2 = images.shape
    assert side == side2 and images.dtype == torch.uint8 and images.is_contiguous()
    bounds, coef = box_tables(side, out_size)
    out = torch.empty((n, 3, out_size, out_size), dtype=torch.float32, device=images.device)
    st = engine.L.vk_preprocess_device(engine.ctx, C.c_void_p(images.data_ptr()), n, side, out_size,
                                       C.c_void_p(bounds.ctypes.data), C.c_void_p(coef.ctypes.data),
                                       coef.shape[1], C.c_float(mean), C.c_float(std), C.c_void_p(out.data_ptr()))
    _capi.check(engine.ctx, st, "vk_preprocess_device")
    return out


def vit(img_size=224, patch=32, dim=1024, depth=24, heads=16, mlp=4096, num_classes=1000):
    """Plain-PyTorch ViT with the shape of timm's vit_large_patch32_224 (the reference's default
    architecture, core/config.py:51-52); random weights."""
    import torch
    from torch import nn

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.n1, self.n2 = nn.LayerNorm(dim, eps=1e-6), nn.LayerNorm(dim, eps=1e-6)
            self.attn = nn.MultiheadAttention(dim, heads, batch_first=True)
            self.mlp = nn.Sequential(nn.Linear(dim, mlp), nn.GELU(), nn.Linear(mlp, dim))

        def forward(self, x):
            h = self.n1(x)
            x = x + self.attn(h, h, h, need_weights=False)[0]
            return x + self.mlp(self.n2(x))

    class ViT(nn.Module):
        def __init__(self):
            super().__init__()
            self.patch = nn.Conv2d(3, dim, patch, patch)
            self.cls = nn.Parameter(torch.zeros(1, 1, dim))
            self.pos = nn.Parameter(torch.zeros(1, (img_size // patch) ** 2 + 1, dim))
            self.blocks = nn.Sequential(*[Block() for _ in range(depth)])
            self.norm = nn.LayerNorm(dim, eps=1e-6)
            self.head = nn.Linear(dim, num_classes)

        def forward(self, x):
            x = self.patch(x).flatten(2).transpose(1, 2)
            x = torch.cat([self.cls.expand(x.shape[0], -1, -1), x], dim=1) + self.pos
            return self.head(self.norm(self.blocks(x))[:, 0])
    return ViT()


def predict(engine, images, model, vocab, threshold=0.7, batch_size=64, multilabel=True, autocast=True):
    """Batched inference like QueryCommand (commands/query.py:283-324): returns (probabilities
    float32 [n, classes] on the host, list of ';'-joined predicted labels or (label, prob))."""
    import torch
    model = model.to(images.device).eval()
    probs = []
    with torch.no_grad():
        for i in range(0, images.shape[0], batch_size):
            x = preprocess(engine, images[i:i + batch_size].contiguous())
            with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
                logits = model(x)
            probs.append((torch.sigmoid(logits) if multilabel else torch.softmax(logits, dim=1)).float().cpu())
    pp = torch.cat(probs)
    if multilabel:
        labels = [";".join(vocab[j] for j in row.nonzero().squeeze(1).tolist()) for row in pp >= threshold]
    else:
        best_p, best_i = torch.max(pp, dim=1)
        labels = [(vocab[i], float(p)) for i, p in zip(best_i.tolist(), best_p.tolist())]
    return pp.numpy(), labels


# ---- the `query` command's output (commands/query.py:225-324) ---------------------------------

COMMON_COLUMNS = ("varKode_image_path", "sample_id", "query_basepairs", "query_kmer_len", "query_mapping",
                  "trained_model_path", "actual_labels", "possible_low_quality", "basefrequency_sd")


def image_metadata(info):
    """(actual_labels, possible_low_quality, basefrequency_sd) from a PNG's text chunks, with the
    reference's getters' behaviour (core/utils.py:71-107): a missing chunk gives NaN, and the
    quality flag is `bool(<string>)`, i.e. True for "False" too."""
    def get(fn):
        try:
            return fn()
        except (AttributeError, TypeError):
            return np.nan
    labels = get(lambda: ";".join(x for x in info.get("varkoderKeywords").split(";")))
    qual = get(lambda: bool(info.get("varkoderLowQualityFlag")))
    freq_sd = get(lambda: float(info.get("varkoderBaseFreqSd")))
    return labels, qual, freq_sd


def predictions_frame(records, probs, vocab, model_path, threshold=0.7, multilabel=True, include_probs=False):
    """The reference's predictions.csv as a DataFrame.  records: one dict per image with
    path, sample, bp, k, mapping, labels, qual, freq_sd; probs float [n, len(vocab)]."""
    import pandas as pd
    common = {
        "varKode_image_path": [r["path"] for r in records],
        "sample_id": [r["sample"] for r in records],
        "query_basepairs": [r["bp"] for r in records],
        "query_kmer_len": [r["k"] for r in records],
        "query_mapping": [r["mapping"] for r in records],
        "trained_model_path": str(model_path),
        "actual_labels": [r["labels"] for r in records],
        "possible_low_quality": [r["qual"] for r in records],
        "basefrequency_sd": [r["freq_sd"] for r in records],
    }
    probs = np.asarray(probs)
    if multilabel:
        predicted = [";".join(vocab[j] for j in np.nonzero(row >= threshold)[0]) for row in probs]
        df = pd.DataFrame({**common, "prediction_type": "Multilabel", "prediction_threshold": threshold,
                           "predicted_labels": predicted})
    else:
        best = probs.argmax(axis=1)
        df = pd.DataFrame({**common, "prediction_type": "Single label", "best_pred_label": [vocab[i] for i in best],
                           "best_pred_prob": probs[np.arange(len(best)), best].tolist()})
    if include_probs:
        df = pd.concat([df, pd.DataFrame(probs, columns=list(vocab))], axis=1)
    return df


def load_model(path):
    """A local TorchScript archive (torch.jit.save) or a pickled torch.nn.Module.  fastai learners
    and HuggingFace hub names, which the reference also accepts (commands/query.py:150-186), need
    packages and a network this build does not assume."""
    import torch
    try:
        return torch.jit.load(str(path), map_location="cpu")
    except RuntimeError:
        return torch.load(str(path), map_location="cpu", weights_only=False)


def read_vocab(path):
    with open(path) as f:
        return [ln.rstrip("\n") for ln in f if ln.strip()]


def probabilities(engine, images, model, batch_size=64, multilabel=True, input_size=224, half=False):
    """float32 [n, classes] on the host for uint8 device images [n, side, side]."""
    import torch
    model = model.to(images.device).eval()
    out = []
    with torch.no_grad():
        for i in range(0, images.shape[0], batch_size):
            x = preprocess(engine, images[i:i + batch_size].contiguous(), out_size=input_size)
            with torch.autocast("cuda", dtype=torch.float16, enabled=half):
                logits = model(x)
            out.append((torch.sigmoid(logits) if multilabel else torch.softmax(logits, dim=1)).float().cpu())
    return torch.cat(out).numpy() if out else np.zeros((0, 0), dtype=np.float32)
