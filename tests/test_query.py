"""Query-side preprocessing: PIL's 8-bit BOX resample restated as coefficient tables,
and the device kernel against PIL + float32 arithmetic."""
import os

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


# ---- the `query` command (commands/query.py:188-324) ------------------------------------------

def _tiny_model(path, classes):
    import torch

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.pool = torch.nn.AdaptiveAvgPool2d(6)
            self.fc = torch.nn.Linear(3 * 36, classes)

        def forward(self, x):
            return self.fc(self.pool(x).flatten(1))
    torch.manual_seed(3)
    m = Tiny()
    with torch.no_grad():
        m.fc.weight.mul_(40.0)
    torch.jit.script(m).save(str(path))
    return m


def _expected_probs(model, arrays, size=224):
    import torch
    from PIL import Image
    xs = []
    for a in arrays:
        r = np.asarray(Image.fromarray(a).resize((size, size), Image.BOX), dtype=np.float32) / 255.0
        xs.append(np.repeat(((r - 0.5) / 0.5)[None], 3, axis=0))
    with torch.no_grad():
        return torch.sigmoid(model(torch.from_numpy(np.stack(xs)))).numpy()


def test_predictions_frame_has_the_reference_columns():
    from varkoder_amd import query as Q
    recs = [dict(path="a.png", sample="a", bp=1000, k=7, mapping="cgr", labels="x;y", qual=True, freq_sd=0.5),
            dict(path="b.png", sample="b", bp=2000, k=7, mapping="cgr", labels=np.nan, qual=np.nan, freq_sd=np.nan)]
    probs = np.array([[0.9, 0.1, 0.7], [0.2, 0.3, 0.1]], dtype=np.float32)
    df = Q.predictions_frame(recs, probs, ["A", "B", "C"], "m.pt", 0.7, True, True)
    assert list(df.columns) == list(Q.COMMON_COLUMNS) + ["prediction_type", "prediction_threshold", "predicted_labels",
                                                          "A", "B", "C"]
    assert list(df["predicted_labels"]) == ["A;C", ""] and set(df["prediction_type"]) == {"Multilabel"}
    df = Q.predictions_frame(recs, probs, ["A", "B", "C"], "m.pt", multilabel=False)
    assert list(df.columns) == list(Q.COMMON_COLUMNS) + ["prediction_type", "best_pred_label", "best_pred_prob"]
    assert list(df["best_pred_label"]) == ["A", "B"] and set(df["prediction_type"]) == {"Single label"}
    # the getters' quirks (core/utils.py:71-107): bool of a non-empty string, NaN for a missing chunk
    assert Q.image_metadata({"varkoderKeywords": "g:x;s:y", "varkoderLowQualityFlag": "False",
                             "varkoderBaseFreqSd": "0.25"}) == ("g:x;s:y", True, 0.25)
    labels, qual, sd = Q.image_metadata({})
    assert np.isnan(labels) and qual is False and np.isnan(sd)


@pytest.mark.gpu
def test_query_command_on_images(tmp_path):
    import glob
    import shutil

    import pandas as pd
    from PIL import Image
    from varkoder_amd import cli
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    indir = tmp_path / "imgs" / "sub"
    indir.mkdir(parents=True)
    paths = []
    for p in sorted(glob.glob(os.path.join(golden, "docs_*+cgr+k7.png"))):
        dst = indir / os.path.basename(p)[len("docs_"):]
        shutil.copyfile(p, dst)
        paths.append(dst)
    vocab = ["kingdom:Animalia", "kingdom:Bacteria", "kingdom:Fungi", "x", "y"]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    model = _tiny_model(tmp_path / "m.pt", len(vocab))
    cli.main(["query", "--images", "-l", str(tmp_path / "m.pt"), "--vocab", str(tmp_path / "vocab.txt"), "-P", "-d", "0.6",
              str(tmp_path / "imgs"), str(tmp_path / "out")])
    df = pd.read_csv(tmp_path / "out" / "predictions.csv")
    want = _expected_probs(model, [np.array(Image.open(p)) for p in paths])
    assert list(df["varKode_image_path"]) == [str(p) for p in paths]
    assert list(df["sample_id"]) == [p.name.split("@")[0] for p in paths]
    assert list(df["query_basepairs"]) == [10_000_000, 200_000_000, 10_000_000] and set(df["query_mapping"]) == {"cgr"}
    assert list(df["actual_labels"].astype(str)) == [Image.open(p).info["varkoderKeywords"] for p in paths]
    got = df[vocab].to_numpy()
    assert np.allclose(got, want, rtol=1e-4, atol=1e-6), np.abs(got - want).max()
    assert list(df["predicted_labels"].fillna("")) == [";".join(v for v, x in zip(vocab, row) if x >= 0.6) for row in want]
    assert 0.02 < want.min() and want.max() < 0.98 or want.std() > 0.05      # the toy model is not saturated everywhere


@pytest.mark.gpu
def test_query_command_from_cleaned_reads(tmp_path):
    import pandas as pd
    from oracle import oracle
    from varkoder_amd import cli, synth
    clean = tmp_path / "int" / "clean_reads"
    clean.mkdir(parents=True)
    blobs = {"q1": synth.sample_fastq(31, 3000, 150, dist=1).tobytes(), "q2": synth.sample_fastq(32, 2000, 150).tobytes()}
    for s, b in blobs.items():
        (clean / f"{s}.fq").write_bytes(b)
    vocab = ["a", "b", "c"]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    model = _tiny_model(tmp_path / "m.pt", 3)
    cli.main(["query", "-l", str(tmp_path / "m.pt"), "--vocab", str(tmp_path / "vocab.txt"), "-k", "7", "-p", "cgr", "-P", "-m",
              str(tmp_path / "int"), str(tmp_path / "out")])
    df = pd.read_csv(tmp_path / "out" / "predictions.csv")
    pix = oracle.cgr_lut(7)
    imgs = [oracle.image(oracle.strand_merge(oracle.count_fastq(blobs[s], 7)[0], 7), 7, pix, 4 ** 7).reshape(128, 128)
            for s in ("q1", "q2")]
    want = _expected_probs(model, imgs)
    assert list(df["sample_id"]) == ["q1", "q2"] and list(df["query_basepairs"]) == [450_000, 300_000]
    assert np.allclose(df[vocab].to_numpy(), want, rtol=1e-4, atol=1e-6)
    assert sorted(p.name for p in (tmp_path / "out" / "query_images").glob("*.png")) == \
        ["q1@00000450K+cgr+k7.png", "q2@00000300K+cgr+k7.png"]


@pytest.mark.gpu
def test_query_command_at_world_two_equals_single_rank(tmp_path):
    """BASELINE config 5's shape on one GPU: `torchrun --nproc-per-node 2 -m varkoder_amd query ...` (both ranks on
    cuda:0) shards the cleaned read files over the ranks -- images on the GPU, batched forward per rank -- and rank 0
    writes the same predictions.csv a single rank writes (rows in input order, same seeds per sample)."""
    import socket
    import subprocess
    import sys

    import pandas as pd
    from varkoder_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    clean = tmp_path / "int" / "clean_reads"
    clean.mkdir(parents=True)
    for i in range(5):
        (clean / f"q{i}.fq").write_bytes(synth.sample_fastq(40 + i, 2000 + 500 * i, 150, dist=i & 1).tobytes())
    vocab = ["a", "b", "c", "d"]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    _tiny_model(tmp_path / "m.pt", len(vocab))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    common = ["query", "-l", str(tmp_path / "m.pt"), "--vocab", str(tmp_path / "vocab.txt"), "-k", "7", "-p", "cgr", "-P",
              str(tmp_path / "int")]
    one = subprocess.run([sys.executable, "-m", "varkoder_amd"] + common + [str(tmp_path / "out1")],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "varkoder_amd"] + common +
                         [str(tmp_path / "out2")], capture_output=True, text=True, timeout=900, cwd=root,
                         env=dict(env, VARKODER_AMD_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert two.returncode == 0, two.stderr[-3000:]
    a, b = pd.read_csv(tmp_path / "out1" / "predictions.csv"), pd.read_csv(tmp_path / "out2" / "predictions.csv")
    assert list(a.columns) == list(b.columns) and list(a["sample_id"]) == list(b["sample_id"]) == [f"q{i}" for i in range(5)]
    assert list(a["query_basepairs"]) == list(b["query_basepairs"])
    assert np.allclose(a[vocab].to_numpy(), b[vocab].to_numpy(), rtol=1e-5, atol=1e-7)
    assert list(a["predicted_labels"].fillna("")) == list(b["predicted_labels"].fillna(""))
