"""Query-side preprocessing: PIL's 8-bit BOX resample restated as coefficient tables,
and the device kernel against PIL + float32 arithmetic."""
import numpy as np
import pytest

from varkoder_amd import query


def pil_reference(img, out=224, mean=0.5, std=0.5):
    from PIL import Image
    r = np.array(Image.fromarray(img).convert("RGB").resize((out, out), resample=Image.Resampling.BOX))
    x = r.astype(np.float32) / np.float32(255.0)
    x = (x - np.float32(mean)) / np.float32(std)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def table_resize(img, out=224):
    side = img.shape[0]
    bounds, coef = query.box_tables(side, out)

    def one_axis(a):  # a: [n_in, m] -> [out, m]
        res = np.empty((out, a.shape[1]), dtype=np.uint8)
        for i in range(out):
            x0, n = bounds[i]
            ss = (1 << 21) + (a[x0:x0 + n].astype(np.int64) * coef[i, :n, None]).sum(axis=0)
            res[i] = np.clip(ss >> 22, 0, 255)
        return res
    tmp = one_axis(img.T).T          # horizontal pass first (PIL's order), 8-bit intermediate
    return one_axis(tmp)


@pytest.mark.parametrize("side", (23, 91, 128, 182, 256, 363, 512))
def test_box_tables_reproduce_pil_resize(side):
    from PIL import Image
    rng = np.random.default_rng(side)
    for trial in range(3):
        img = rng.integers(0, 256, (side, side), dtype=np.uint8)
        if trial == 2:
            img = (img // 128 * 255).astype(np.uint8)       # hard edges: rounding ties
        want = np.array(Image.fromarray(img).resize((224, 224), resample=Image.Resampling.BOX))
        assert np.array_equal(table_resize(img), want)


@pytest.mark.gpu
@pytest.mark.parametrize("side", (91, 128, 256, 512))
def test_preprocess_kernel_equals_pil_pipeline(engines, side):
    import torch
    eng = engines(7)
    rng = np.random.default_rng(side)
    imgs = rng.integers(0, 256, (5, side, side), dtype=np.uint8)
    x = query.preprocess(eng, torch.from_numpy(imgs).cuda())
    got = x.cpu().numpy()
    assert got.shape == (5, 3, 224, 224) and got.dtype == np.float32
    for i in range(5):
        assert np.array_equal(got[i], pil_reference(imgs[i])), i      # bit-identical float32


@pytest.mark.gpu
def test_predict_runs_a_batched_forward(engines):
    import torch
    eng = engines(7)
    torch.manual_seed(0)
    model = query.vit(dim=64, depth=2, heads=4, mlp=128, num_classes=5)
    imgs = torch.randint(0, 256, (7, 128, 128), dtype=torch.uint8, device="cuda")
    vocab = [f"taxon{i}" for i in range(5)]
    pp, labels = query.predict(eng, imgs, model, vocab, threshold=0.5, batch_size=4)
    assert pp.shape == (7, 5) and len(labels) == 7 and np.isfinite(pp).all()
    for row, lab in zip(pp, labels):
        assert lab == ";".join(v for v, p in zip(vocab, row) if p >= 0.5)
