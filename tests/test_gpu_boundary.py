"""The drop-in boundary on the GPU: count_kmers()/make_image() with the reference's
signatures, file naming, PNG metadata and error behaviour (commands/image.py:727-936)."""
import hashlib

import numpy as np
import pytest

from oracle import oracle
from varkoder_amd import image, synth
from varkoder_amd.mapping import get_kmer_mapping, pixel_lut, side

pytestmark = pytest.mark.gpu


def _write_fastq(path, sample, reads, gz=False):
    data = synth.sample_fastq(sample, reads, 150, dist=1).tobytes()
    if gz:
        import gzip
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        path.write_bytes(data)
    return data


@pytest.mark.parametrize("k,mapping,gz", [(7, "cgr", True), (7, "varKode", False), (5, "cgr", False)])
def test_count_then_image_like_run_clean2img(tmp_path, k, mapping, gz):
    from PIL import Image
    fq = tmp_path / ("SRR1@00000500K.fq.gz" if gz else "SRR1@00000500K.fq")
    data = _write_fastq(fq, 9, 3500, gz)
    counts_d, images_d = tmp_path / f"{k}mer_counts", tmp_path / "images"

    st = image.count_kmers(fq, counts_d, threads=1, k=k)
    assert list(st.keys()) == [f"{k}mer_counting_time"]
    h5 = counts_d / f"SRR1@00000500K+k{k}.fq.h5"
    assert h5.is_file()
    assert image.count_kmers(fq, counts_d, k=k) == {}            # exists, no overwrite -> skipped

    kk, hist = image.read_counts(h5)
    want_h = oracle.count_fastq(data, k)[0]
    assert kk == k and np.array_equal(hist, want_h)

    kmap = get_kmer_mapping(k, mapping)
    st = image.make_image(h5, images_d, kmap, labels=["genus:Bembidion", "sp:x"], base_sd=0.0123,
                          mapping_code=mapping)
    assert list(st.keys()) == [f"k{k}_img_time"]
    png = images_d / f"SRR1@00000500K+{mapping}+k{k}.png"
    assert png.is_file()
    assert image.make_image(h5, images_d, kmap, mapping_code=mapping) == {}   # skipped

    im = Image.open(png)
    assert im.mode == "L"
    n = side(k, mapping)
    want_i = oracle.image(oracle.strand_merge(want_h, k), k, pixel_lut(k, mapping), n * n).reshape(n, n)
    assert np.array_equal(np.array(im), want_i)
    info = {a: b for a, b in im.info.items() if a.startswith("varkoder")}
    assert list(info.keys()) == ["varkoderKeywords", "varkoderBaseFreqSd", "varkoderLowQualityFlag",
                                 "varkoderMapping"]
    assert info == {"varkoderKeywords": "genus:Bembidion;sp:x", "varkoderBaseFreqSd": "0.0123",
                    "varkoderLowQualityFlag": "True", "varkoderMapping": mapping}


def test_png_metadata_equals_reference_cases(tmp_path, manifest):
    """Text chunks, file name and stats key as the reference's make_image wrote them
    (tests/golden/manifest.json, png_metadata_cases; produced by oracle/gen_golden.py)."""
    from PIL import Image
    import vectors
    kmap = get_kmer_mapping(5, "cgr")
    for case in manifest["png_metadata_cases"]:
        h5 = tmp_path / f"{case['sample']}@00010000K+k5.fq.h5"
        image.write_counts(h5, 5, vectors.fwd_hist(5, "heavy"))
        st = image.make_image(h5, tmp_path / "out", kmap, overwrite=True, labels=case["labels"],
                              base_sd=case["base_sd"], mapping_code="cgr")
        assert list(st.keys()) == case["stats_keys"]
        png = tmp_path / "out" / case["filename"]
        assert png.is_file()
        im = Image.open(png)
        info = {a: b for a, b in im.info.items() if a.startswith("varkoder")}
        assert info == case["info"]
        got = hashlib.sha256(np.array(im).astype(np.uint8).tobytes()).hexdigest()
        assert got == manifest["image_cases"]["k5_cgr_heavy"]["sha256"]


def test_subfolder_levels_and_errors(tmp_path):
    import pandas as pd
    kmap = get_kmer_mapping(5, "varKode")
    h5 = tmp_path / "S@00000010K+k5.fq.h5"
    image.write_counts(h5, 5, np.ones(4 ** 5, dtype=np.uint32))
    image.make_image(h5, tmp_path / "img", kmap, subfolder_levels=2, mapping_code="varKode")
    name = "S@00000010K+varKode+k5.png"
    hsh = list(hashlib.md5(name.encode("UTF-8")).hexdigest())
    assert (tmp_path / "img" / hsh.pop() / hsh.pop() / name).is_file()      # image.py:850-854
    image.write_counts(h5, 5, np.zeros(4 ** 5, dtype=np.uint32))
    with pytest.raises(pd.errors.EmptyDataError):                            # empty dsk2ascii dump
        image.make_image(h5, tmp_path / "img2", kmap, mapping_code="varKode")
    bad = tmp_path / "B@00000010K.fq"
    bad.write_bytes(b"@r\nACGTACGTAC\n+\n")                                  # truncated record
    with pytest.raises(RuntimeError):
        image.count_kmers(bad, tmp_path / "c", k=5)


def test_count_kmers_takes_an_empty_text_gz_and_rejects_a_damaged_one(tmp_path):
    """The function entry and the dsk shim (tests/test_shims.py) agree on the same inputs: a valid .gz whose text is
    empty is a sample without k-mers -- dsk exits 0 on it, so `check=True` passes (commands/image.py:791-796), and the
    empty dsk2ascii dump only fails later, in make_image's read_csv (:897-899) -- while a damaged .gz makes dsk fail."""
    import gzip
    import pandas as pd
    k = 7
    empty = tmp_path / "E@00000001K.fq.gz"
    empty.write_bytes(gzip.compress(b""))
    st = image.count_kmers(empty, tmp_path / "c", k=k)
    assert list(st.keys()) == [f"{k}mer_counting_time"]
    kk, hist = image.read_counts(tmp_path / "c" / f"E@00000001K+k{k}.fq.h5")
    assert kk == k and not hist.any()
    with pytest.raises(pd.errors.EmptyDataError):       # the reference's IMAGE FAIL path (image.py:1111-1118 catches ParserError's family)
        image.make_image(tmp_path / "c" / f"E@00000001K+k{k}.fq.h5", tmp_path / "i", get_kmer_mapping(k, "cgr"), mapping_code="cgr")
    data = synth.sample_fastq(3, 2000, 150, dist=1).tobytes()
    blob = bytearray(gzip.compress(data))
    blob[len(blob) // 2] ^= 0xFF
    bad = tmp_path / "B@00000001K.fq.gz"
    bad.write_bytes(bytes(blob))
    with pytest.raises(RuntimeError, match="not a readable"):
        image.count_kmers(bad, tmp_path / "c", k=k)
    assert not (tmp_path / "c" / f"B@00000001K+k{k}.fq.h5").exists()
    empty_plain = tmp_path / "P@00000001K.fq"            # a zero-byte plain file: the same empty table
    empty_plain.write_bytes(b"")
    assert list(image.count_kmers(empty_plain, tmp_path / "c", k=k).keys()) == [f"{k}mer_counting_time"]


def test_batched_pipeline_writes_reference_named_pngs(tmp_path):
    """pipeline.fastqs_to_images: many split FASTQs per launch, PNGs + stats keys as steps D+E
    of run_clean2img (image.py:1054-1127); pixels equal the oracle's."""
    from PIL import Image
    from varkoder_amd import pipeline
    files, datas = [], {}
    for s, reads in enumerate((1200, 800, 2500, 400)):
        f = tmp_path / f"samp{s}@{reads * 150 // 1000:08d}K.fq"
        datas[f.name] = _write_fastq(f, 40 + s, reads)
        files.append(f)
    bad = tmp_path / "broken@00000001K.fq"
    bad.write_bytes(b"@r\nACGTACGTACGT\n+\n")
    files.append(bad)
    stats = pipeline.fastqs_to_images(files, tmp_path / "images", k=7, mapping_code="varKode",
                                      labels={"samp1": ["t:a", "t:b"]}, batch_bytes=600000)
    assert stats["broken@00000001K"] == {"failed_step": "image"}
    n = side(7, "varKode")
    for f in files[:-1]:
        key = f.name[:-3]
        assert list(stats[key].keys()) == ["7mer_counting_time", "k7_img_time"]
        png = tmp_path / "images" / f"{key}+varKode+k7.png"
        im = Image.open(png)
        want, _, st = oracle.fastq_to_image(datas[f.name], 7, pixel_lut(7, "varKode"), n * n)
        assert st == 0 and np.array_equal(np.array(im).ravel(), want)
        assert im.info["varkoderKeywords"] == ("t:a;t:b" if key.startswith("samp1") else "")
    # a second run skips what exists (checkpoint/resume by file existence, image.py:857-859)
    again = pipeline.fastqs_to_images(files[:-1], tmp_path / "images", k=7, mapping_code="varKode")
    assert again == {}


def test_plain_files_from_the_page_cache_equal_the_staged_route(tmp_path, monkeypatch):
    """Plain-text files reach the GPU by DMA from their page-cache pages (engine.stage_files maps them,
    vk_upload_mapped registers and copies them); VARKODER_AMD_MMAP=0 (and any rank with 16 I/O threads or more) reads them into the pinned staging buffer.
    Same text in HBM either way -- sizes that are no multiple of 16 and of the page size, an empty file, a gzip
    file in the same batch -- and the same PNGs from the pipeline."""
    from varkoder_amd import engine as engine_mod, pipeline
    from varkoder_amd.engine import ImageEngine
    files, datas = [], []
    for s, reads in enumerate((700, 1, 2048, 333)):
        f = tmp_path / f"m{s}@{reads * 150 // 1000:08d}K.fq"
        datas.append(_write_fastq(f, 60 + s, reads))
        files.append(f)
    empty = tmp_path / "e@00000000K.fq"
    empty.write_bytes(b"")
    gz = tmp_path / "z@00000105K.fq.gz"
    datas_gz = _write_fastq(gz, 70, 700, gz=True)
    batch = files[:2] + [empty, gz] + files[2:]
    want = datas[:2] + [b"", datas_gz] + datas[2:]
    eng = ImageEngine(k=7, mapping="cgr")
    got = {}
    for mapped in (True, False):
        monkeypatch.setattr(engine_mod, "USE_MAPPED_UPLOAD", mapped)
        st = eng.stage_files(batch)
        assert set(st["mapped"]) >= ({0, 1, 4, 5} if mapped else set()) and (mapped or not st["mapped"])   # (+ the gzip file)
        dev, offs, lens = eng.upload_staged(st)
        host = dev.cpu().numpy()
        for i, w in enumerate(want):
            o, n = int(offs[i]), int(lens[i])
            assert n == len(w) and bytes(host[o:o + n]) == w, (mapped, i)
            assert not host[o + n:(o + n + 15) // 16 * 16].any()       # zero up to the 16-byte rounded end
        out = tmp_path / ("img%d" % mapped)
        stats = pipeline.fastqs_to_images(batch, out, k=7, mapping_code="cgr", engine=eng, batch_bytes=300000)
        assert stats["e@00000000K"] == {"failed_step": "image"}
        got[mapped] = {p.name: hashlib.sha256(p.read_bytes()).hexdigest() for p in sorted(out.glob("*.png"))}
    eng.close()
    assert len(got[True]) == 5 and got[True] == got[False]


def test_cli_image_on_an_intermediate_folder(tmp_path):
    """`python -m varkoder_amd image INT -k 7 -p cgr -o OUT -f stats.csv -t`: same flags as the
    reference CLI (cli.py:69-166), entering at step D on a folder of split FASTQs."""
    import pandas as pd
    from PIL import Image
    from varkoder_amd import cli
    split = tmp_path / "int" / "split_fastqs"
    split.mkdir(parents=True)
    datas = {}
    for name, s, reads in (("tax1_A@00000150K.fq", 60, 1000), ("tax1_A@00000075K.fq", 61, 500),
                           ("tax2_B@00000150K.fq.gz", 62, 1000)):
        datas[name] = _write_fastq(split / name, s, reads, gz=name.endswith(".gz"))
    (tmp_path / "labels.csv").write_text("sample,labels\ntax1_A,genus:A;family:F\ntax2_B,genus:B\n")
    out, stats = tmp_path / "images", tmp_path / "stats.csv"
    cli.main(["image", str(tmp_path / "int"), "-k", "7", "-p", "cgr", "-o", str(out), "-f", str(stats), "-t",
              "--labels-csv", str(tmp_path / "labels.csv"), "-n", "2"])
    for name, data in datas.items():
        stem = name.split(".fq")[0]
        im = Image.open(out / f"{stem}+cgr+k7.png")
        want, _, st = oracle.fastq_to_image(data, 7, pixel_lut(7, "cgr"), 128 * 128)
        assert st == 0 and np.array_equal(np.array(im).ravel(), want)
        assert im.info["varkoderMapping"] == "cgr"
    assert Image.open(out / "tax1_A@00000150K+cgr+k7.png").info["varkoderKeywords"] == "genus:A;family:F"
    df = pd.read_csv(stats)
    assert list(df["sample"]) == ["tax1_A", "tax2_B"]
    assert {"7mer_counting_time", "k7_img_time"} <= set(df.columns)
    lt = pd.read_csv(out / "labels.csv")
    assert list(lt.columns) == ["sample", "labels", "possible_low_quality"]
    with pytest.raises(Exception, match="Output directory exists"):
        cli.main(["image", str(tmp_path / "int"), "-o", str(out), "-f", str(stats)])
    with pytest.raises(ValueError, match="between 5 and 9"):
        cli.main(["image", str(tmp_path / "int"), "-k", "4", "-o", str(tmp_path / "o2")])


def test_abi_argument_errors(engines):
    """Status codes, never crashes: unaligned sample offsets, missing mapping, bad k."""
    import ctypes as C
    import torch
    from varkoder_amd import _capi
    eng = engines(7)
    L = eng.L
    fq = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    hist = torch.zeros(4 ** 7, dtype=torch.int32, device="cuda")
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    offs, lens = (C.c_uint64 * 1)(8), (C.c_uint64 * 1)(100)          # offset not a multiple of 16
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    assert L.vk_count_device(eng.ctx, vp(fq), offs, lens, 1, 7, 0, vp(hist), vp(st)) == _capi.VK_EINVAL
    offs[0] = 0
    assert L.vk_count_device(eng.ctx, vp(fq), offs, lens, 1, 4, 0, vp(hist), vp(st)) == _capi.VK_EINVAL
    assert L.vk_count_device(eng.ctx, vp(fq), offs, lens, 0, 7, 0, vp(hist), vp(st)) == _capi.VK_OK   # empty batch
    ctx = C.c_void_p()
    assert L.vk_ctx_create(0, None, 1, C.byref(ctx)) == _capi.VK_OK
    img = torch.zeros(128 * 128, dtype=torch.uint8, device="cuda")
    assert L.vk_image_device(ctx, vp(hist), 1, 7, vp(img)) == _capi.VK_ENOMAP        # no vk_set_mapping yet
    assert L.vk_set_mapping(ctx, 7, None, 100) == _capi.VK_EINVAL                    # cgr needs npix = 4^k
    L.vk_ctx_destroy(ctx)
    with pytest.raises(_capi.VkError, match="invalid argument"):
        _capi.check(eng.ctx, _capi.VK_EINVAL, "demo")


def test_bench_script_runs_end_to_end(tmp_path):
    """bench.py with a tiny workload: the JSON contract fields are all there."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--samples", "16", "--pool", "8", "--reads", "20000", "--cpu-seconds", "0.5", "--e2e-files", "6",
                          "--e2e-reads", "5000", "--config4-samples", "6", "--config4-steps", "2", "--realistic-pool", "8", "--realistic-steps", "2", "--ladder-samples", "3",
                          "--query-samples", "12", "--query-batch", "8", "--query-steps", "2"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Gbases/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["vs_baseline"] is None
    assert d["bad_status_samples"] == 0 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert "workload" in d["config"]
    assert r["kernel"] == "vk_count_dense_kernel" and "vk_count_dense_kernel(+check)" in d["kernel_ms"]
    # the HBM traffic of the count launch is measured in the run itself (the script again, under rocprofv3 --pmc).  (Here the 8
    # distinct samples of 6.4 MB are read twice each and stay in the caches: at least the distinct text, no more than all of it.)
    # (Where rocprofv3 cannot run a child of this process -- a harness that preloads its own library, say -- the line says so in
    # `traffic_live_error` and keeps the committed profile's figure: that is the documented fallback, not a failure of the bench.)
    if "traffic_live_error" not in r:
        assert r["traffic_source"].startswith("measured_in_this_run")
        assert 0.45 * r["algorithmic_bytes_per_launch"] <= r["traffic"] <= 1.5 * r["algorithmic_bytes_per_launch"]
    else:
        assert r["traffic_source"].startswith("from_profile_file")
    e = d["end_to_end"]
    for leg in ("plain_text", "fq_gz"):
        assert e[leg]["all_files_ok"] and e[leg]["pngs"] == 6 and e[leg]["gbases_per_s"] > 0
        assert len(e[leg]["passes_s"]) == 3 and e[leg]["seconds"] == sorted(e[leg]["passes_s"])[1]     # the median
        assert {"stage_wait_s", "upload_s", "inflate_s", "kernels_s", "png_tail_s"} <= set(e[leg]["where_the_median_pass_went_s"])
    assert e["fq_gz"]["file_bytes"] < e["plain_text"]["file_bytes"] and e["gz_pngs_identical_to_plain"]
    c4 = d["config4"]                                   # BASELINE configs[3] as a side leg: k=9 cgr, both distributions
    assert c4["k"] == 9 and c4["samples"] == 6
    for leg in ("dist0", "dist1", "dist2"):
        assert c4[leg]["bad_status_samples"] == 0 and c4[leg]["count_ms"] > 0
        rr = c4[leg]["roofline"]
        assert rr["bound"] == "hbm" and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-9
        if "traffic_live_error" not in rr and "traffic_live_error" not in r:
            assert rr["traffic_source"].startswith("measured_in_this_run") and rr["traffic"] >= 0.9 * rr["algorithmic_bytes_per_launch"]
            assert rr["traffic_from_profile_file"] is None      # (the committed profile is of another configuration)
        else:
            assert rr["traffic_source"].startswith("from_profile_file")
    assert c4["dist2"]["fastq_bytes"] != c4["dist0"]["fastq_bytes"]
    # the line checks itself: the first batch entries' histograms and images against the oracle's
    assert d["verified_samples"] == 4 and d["cpu_baseline"]["verified_samples"] == 4 and "verify_failed" not in d
    qy = d["query"]                                     # BASELINE configs[4] on one GPU: images -> input transform -> forward
    assert qy["samples"] == 12 and qy["verified_preprocess_images"] == 4 and qy["bad_status_samples"] == 0
    assert qy["images_per_s"] > 0 and qy["preprocess_ms"] > 0 and qy["forward_ms"] > 0 and 0 < qy["forward_share"] < 1
    assert abs(qy["preprocess_gb_per_s"] - qy["preprocess_bytes"] / (qy["preprocess_ms"] * 1e-3) / 1e9) < 1e-6
    rl = d["realistic"]                                 # reads of the lengths fastp writes (synth.py dist 2)
    for leg in ("dense", "classic"):
        assert rl[leg]["bad_status_samples"] == 0 and rl[leg]["count_ms"] > 0
    assert 0 < rl["dense"]["general_piece_fraction"] <= 1 and rl["dense"]["pieces"] > 0
    assert d["ladder"]["samples"] == 3 and d["ladder"]["steps"] >= 3 and d["ladder"]["ms"] > 0 and d["ladder"]["failed_samples"] == 0


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`bench.py --gpus 2` with no launcher: the script starts its two ranks (both on cuda:0 here,
    gloo for the control plane) and the line it prints says so; each rank runs half of the batch."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--all-on-device0",
                          "--backend", "gloo", "--steps", "2", "--warmup", "1", "--total-samples", "24", "--pool", "6",
                          "--reads", "20000", "--e2e-files-per-rank", "3", "--e2e-reads", "5000", "--ladder-shard-samples", "2"],
                         capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["bad_status_samples"] == 0
    assert d["config"]["samples_per_gpu"] == 12
    assert d["config"]["process_group"] == {"backend": "gloo", "world_size": 2}
    assert "configs[2]" in d["config"]["workload"] and "cpu_baseline" not in d and "config4" not in d
    assert len(d["ms_per_step_by_rank"]["all"]) == 2 and d["ms_per_step_by_rank"]["max"] <= d["ms_per_step"] * 1.0001
    e = d["end_to_end"]                                  # every rank runs the file pipeline on its own files at once
    assert e["all_files_ok"] and e["files_per_rank"] == 3 and e["gbases_per_s"] > 0
    assert all(len(p) == 2 for p in e["passes_s_by_rank"]) and len(e["passes_s_by_rank"]) == 3
    ls = d["ladder_shard"]                                # units of unequal size (the reference's six-rung ladder), sharded by size
    assert ls["units"] == 12 and len(ls["bytes_by_rank"]) == 2 and len(ls["ms_by_rank_median_pass"]) == 2
    assert ls["static_deal_max_over_mean_bytes"] <= 1.05 < ls["round_robin_max_over_mean_bytes"] and ls["bad_status_units_this_rank"] == 0
    # round 6: the last tenth of the bytes is pulled from a shared cursor by whichever rank is free -- every unit still once
    assert ls["head_units"] + ls["tail_units"] == 12 and ls["tail_units"] >= 1
    assert sum(ls["bytes_by_rank"]) == sum(ls["static_deal_bytes_by_rank"]) and ls["max_over_mean_ms"] >= 1.0
    # a launcher that disagrees with --gpus is an error, not a silent one-rank run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                         capture_output=True, text=True, timeout=300, cwd=root, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert bad.returncode == 2 and "WORLD_SIZE" in bad.stderr


def test_c_abi_from_plain_c(tmp_path):
    """examples/fastq_to_pgm.c: the shared library used from C without Python or PyTorch in the
    process; its image equals the oracle's."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    exe = tmp_path / "fastq_to_pgm"
    libdir = os.path.join(root, "varkoder_amd")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "fastq_to_pgm.c"), "-o", str(exe), "-L", libdir,
                           "-l:libvkimg_hip.so", f"-Wl,-rpath,{libdir}"])
    fq = tmp_path / "s.fq"
    data = _write_fastq(fq, 77, 5000)
    out = tmp_path / "s.pgm"
    res = subprocess.run([str(exe), str(fq), "7", str(out)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    raw = out.read_bytes()
    assert raw.startswith(b"P5\n128 128\n255\n")
    got = np.frombuffer(raw[len(b"P5\n128 128\n255\n"):], dtype=np.uint8)
    want, nwin, st = oracle.fastq_to_image(data, 7, pixel_lut(7, "cgr"), 128 * 128)
    assert st == 0 and np.array_equal(got, want)
    assert f"{nwin} k-mer windows" in res.stdout


def test_cli_image_at_world_two_equals_single_rank(tmp_path):
    """The product entry under a launcher: `torchrun --nproc-per-node 2 -m varkoder_amd image ...` (both
    ranks on cuda:0 here, started by torch.distributed.run BEFORE anything touches the GPU) writes the
    same PNG files and the same merged stats.csv rows as a single-rank run -- files are dealt by size
    (shard.shard_by_size), rank 0 merges the per-sample stats (reference: commands/image.py:1281-1284, 1144-1170)."""
    import os
    import socket
    import subprocess
    import sys

    import pandas as pd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    split = tmp_path / "int" / "split_fastqs"
    split.mkdir(parents=True)
    for i in range(7):
        _write_fastq(split / f"tax{i}_S@{150 * (1 + i % 3):08d}K.fq{'.gz' if i % 2 else ''}", 70 + i, 1000 * (1 + i % 3),
                     gz=bool(i % 2))
    (split / "broken_Z@00000150K.fq").write_bytes(b"this is not a FASTQ file\n" * 50)   # one bad sample never kills the run
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT",
                                                            "LOCAL_WORLD_SIZE")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    common = ["image", str(tmp_path / "int"), "-k", "7", "-p", "varKode", "-n", "2"]
    one = subprocess.run([sys.executable, "-m", "varkoder_amd"] + common +
                         ["-o", str(tmp_path / "img1"), "-f", str(tmp_path / "stats1.csv")],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "varkoder_amd"] + common +
                         ["-o", str(tmp_path / "img2"), "-f", str(tmp_path / "stats2.csv")],
                         capture_output=True, text=True, timeout=900, cwd=root,
                         env=dict(env, VARKODER_AMD_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert two.returncode == 0, two.stderr[-3000:]
    a = sorted(p.name for p in (tmp_path / "img1").rglob("*.png"))
    b = sorted(p.name for p in (tmp_path / "img2").rglob("*.png"))
    assert a == b and len(a) == 7
    for name in a:
        assert next((tmp_path / "img1").rglob(name)).read_bytes() == next((tmp_path / "img2").rglob(name)).read_bytes()
    s1, s2 = pd.read_csv(tmp_path / "stats1.csv"), pd.read_csv(tmp_path / "stats2.csv")
    assert list(s1["sample"]) == list(s2["sample"]) and len(s1) == 8
    assert list(s1.columns) == list(s2.columns)
    assert list(s1["failed_step"].fillna("")) == list(s2["failed_step"].fillna(""))
    assert s2.set_index("sample").loc["broken_Z", "failed_step"] == "image"
    timed = [c for c in s2.columns if c.endswith("_time")]
    assert timed and (s2.set_index("sample").drop("broken_Z")[timed] > 0).all().all()


def _torchrun_image(root, env, port, args, timeout=900):
    import subprocess
    import sys
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "varkoder_amd"] + args,
                          capture_output=True, text=True, timeout=timeout, cwd=root,
                          env=dict(env, VARKODER_AMD_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0"))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _launcher_env(root):
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT",
                                                            "LOCAL_WORLD_SIZE")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    return env


@pytest.mark.gpu
def test_cli_image_at_world_two_sums_a_samples_ladder_over_the_ranks(tmp_path):
    """Ladder-shaped input -- four rungs per sample, as split_fastq leaves them (commands/image.py:682-708): dealt by size,
    a sample's rungs land on BOTH ranks.  The PNG bytes equal a single-rank run's, and a sample's `<k>mer_counting_time`
    / `k<k>_img_time` in stats.csv is the sum over ALL its files on all ranks (the reference accumulates over a sample's
    files, commands/image.py:1078) -- until round 6 the last rank's share overwrote the others'."""
    import json
    import os
    import subprocess
    import sys

    import pandas as pd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    split = tmp_path / "int" / "split_fastqs"
    split.mkdir(parents=True)
    rungs = [150, 300, 750, 1500]   # Kbp names; reads per file in proportion
    for i in range(3):
        for kbp in rungs:
            _write_fastq(split / f"tax{i}_S@{kbp:08d}K.fq", 200 + 10 * i + kbp % 7, kbp * 1000 // 150)
    env = _launcher_env(root)
    common = ["image", str(tmp_path / "int"), "-k", "7", "-p", "cgr", "-n", "2"]
    one = subprocess.run([sys.executable, "-m", "varkoder_amd"] + common +
                         ["-o", str(tmp_path / "img1"), "-f", str(tmp_path / "stats1.csv")],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    two = _torchrun_image(root, dict(env, VARKODER_AMD_PER_FILE_STATS=str(tmp_path / "perfile")), _free_port(),
                          common + ["-o", str(tmp_path / "img2"), "-f", str(tmp_path / "stats2.csv")])
    assert two.returncode == 0, two.stderr[-3000:]
    a = sorted(p.name for p in (tmp_path / "img1").rglob("*.png"))
    b = sorted(p.name for p in (tmp_path / "img2").rglob("*.png"))
    assert a == b and len(a) == 12
    for name in a:
        assert next((tmp_path / "img1").rglob(name)).read_bytes() == next((tmp_path / "img2").rglob(name)).read_bytes()
    parts = [json.load(open(str(tmp_path / "perfile") + f".rank{r}.json")) for r in range(2)]
    assert all(parts) and not set(parts[0]) & set(parts[1]) and len(parts[0]) + len(parts[1]) == 12
    # the rungs of at least one sample were split over the ranks (LPT over four sizes x three samples on two ranks)
    assert any({k.split("@")[0] for k in parts[0]} & {k.split("@")[0] for k in parts[1]})
    s2 = pd.read_csv(tmp_path / "stats2.csv").set_index("sample")
    for col in ("7mer_counting_time", "k7_img_time"):
        for sample in (f"tax{i}_S" for i in range(3)):
            want = sum(v[col] for part in parts for k, v in part.items() if k.split("@")[0] == sample)
            assert abs(float(s2.loc[sample, col]) - want) <= 1e-9 * max(1.0, want), (sample, col)


@pytest.mark.gpu
def test_cli_image_with_a_failing_rank_ends_at_once(tmp_path):
    """One rank of a two-rank `image` job raises before its share (VARKODER_AMD_FAULT: the stand-in for a device out of
    memory or a full disk): the job ends within seconds with a non-zero exit code -- the failing rank still goes to the
    gather and the barrier -- instead of hanging its peer in gather_object until the gloo timeout (30 minutes); the
    other rank's images and stats rows are there."""
    import os
    import time

    import pandas as pd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    split = tmp_path / "int" / "split_fastqs"
    split.mkdir(parents=True)
    for i in range(4):
        _write_fastq(split / f"tax{i}_S@{150 * (1 + i):08d}K.fq", 300 + i, 1000 * (1 + i))
    env = _launcher_env(root)
    t0 = time.time()
    two = _torchrun_image(root, dict(env, VARKODER_AMD_FAULT="rank1"), _free_port(),
                          ["image", str(tmp_path / "int"), "-k", "7", "-p", "cgr", "-n", "2", "-o", str(tmp_path / "img"),
                           "-f", str(tmp_path / "stats.csv")], timeout=600)
    took = time.time() - t0
    assert two.returncode != 0
    assert "injected fault on rank 1" in two.stderr
    assert took < 240, took                        # (start-up of two ranks + one rank's share; the gloo timeout is 1800 s)
    pngs = sorted(p.name for p in (tmp_path / "img").rglob("*.png"))
    assert 1 <= len(pngs) < 4                      # rank 0's share was made
    st = pd.read_csv(tmp_path / "stats.csv")
    assert 1 <= len(st) < 4
