"""Host-side mirror of the reference interface: mappings, file containers, naming, and the
rule that the product path never falls back to the CPU."""
import os

import numpy as np
import pytest

from varkoder_amd import _capi, config, image, mapping


def test_constants_match_reference_config():
    # varKoder/core/config.py:18-24, 33-34
    assert (config.LABEL_SAMPLE_SEP, config.LABELS_SEP, config.BP_KMER_SEP, config.SAMPLE_BP_SEP) == ("+", ";", "+", "@")
    assert config.QUAL_THRESH == 0.01
    assert config.DEFAULT_KMER_SIZE == 7 and config.DEFAULT_KMER_MAPPING == "cgr"
    assert config.MAPPING_CHOICES == ["varKode", "cgr"]


def test_get_kmer_mapping_shapes_and_errors():
    df = mapping.get_kmer_mapping(5, "varKode")
    assert df.shape == (1024, 2) and list(df.columns) == ["x", "y"] and df.index.name == "kmer"
    assert int(df["x"].max()) + 1 == 23 and int(df["y"].max()) + 1 == 23
    cg = mapping.get_kmer_mapping(6, "cgr")
    assert cg.shape == (2 * 4 ** 6, 2) and int(cg["x"].max()) == 63
    # second half: reverse-complement spelling at the original's coordinates (utils.py:201-210)
    assert cg.index[4 ** 6] == "T" * 6 and tuple(cg.iloc[4 ** 6]) == tuple(cg.iloc[0])
    with pytest.raises(Exception, match='method must be "varKode" or "cgr"'):
        mapping.get_kmer_mapping(7, "nope")
    with pytest.raises(ValueError, match="between 5 and 9"):
        mapping.get_kmer_mapping(4, "cgr")


@pytest.mark.parametrize("k,method", [(5, "cgr"), (6, "varKode"), (7, "cgr")])
def test_lut_roundtrip_through_dataframe(k, method):
    df = mapping.get_kmer_mapping(k, method)
    want = mapping.pixel_lut(k, method)
    k1, lut, npix = mapping.lut_from_dataframe(df)
    assert k1 == k and np.array_equal(lut, want) and npix == mapping.side(k, method) ** 2
    plain = df.copy()
    plain.attrs = {}
    k2, lut2, npix2 = mapping.lut_from_dataframe(plain)      # as if it came from the reference
    assert k2 == k and np.array_equal(lut2, want) and npix2 == npix


def test_counts_container_roundtrip_and_errors(tmp_path):
    import pandas as pd
    h = np.arange(4 ** 5, dtype=np.uint32)
    p = tmp_path / "S@00010000K+k5.fq.h5"
    image.write_counts(p, 5, h)
    k, back = image.read_counts(p)
    assert k == 5 and np.array_equal(back, h)
    p.write_bytes(p.read_bytes()[:-3])
    with pytest.raises(pd.errors.ParserError):
        image.read_counts(p)
    p.write_bytes(b"ACGTA 3\n")
    with pytest.raises(pd.errors.ParserError):
        image.read_counts(p)


def test_gzip_and_plain_fastq_read_the_same(tmp_path):
    import gzip
    data = b"@r\nACGTACGTAC\n+\nIIIIIIIIII\n"
    (tmp_path / "a.fq").write_bytes(data)
    with gzip.open(tmp_path / "a.fq.gz", "wb") as f:
        f.write(data)
    assert image.read_fastq_bytes(tmp_path / "a.fq") == data
    assert image.read_fastq_bytes(tmp_path / "a.fq.gz") == data


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="checks the behaviour on a GPU-less host")
def test_product_path_fails_loudly_without_gpu(tmp_path):
    """No CPU fallback: without a GPU the mirror functions raise instead of computing."""
    fq = tmp_path / "S@00010000K.fq"
    fq.write_bytes(b"@r\nACGTACGTAC\n+\nIIIIIIIIII\n")
    with pytest.raises(_capi.VkError):
        image.count_kmers(fq, tmp_path / "counts", k=5)
    image.write_counts(tmp_path / "S@00010000K+k5.fq.h5", 5, np.ones(4 ** 5, dtype=np.uint32))
    with pytest.raises(_capi.VkError):
        image.make_image(tmp_path / "S@00010000K+k5.fq.h5", tmp_path / "img",
                         mapping.get_kmer_mapping(5, "cgr"), mapping_code="cgr")


def test_missing_library_is_an_error(monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", os.path.join(os.path.dirname(_capi.LIB_PATH), "nope.so"))
    with pytest.raises(_capi.VkError, match="no CPU fallback"):
        _capi.lib()


def test_dsk_text_dump_roundtrip():
    """dsk2ascii-style text: one line per canonical class, parses back to the class
    totals whichever spelling convention is printed."""
    import vectors
    from varkoder_amd import formats
    for k in (5, 6):
        fwd = vectors.fwd_hist(k, "heavy")
        want = vectors.class_totals(fwd, k).astype(np.uint64)
        for conv in ("gatb", "lex"):
            text = formats.dsk_text(fwd, k, conv)
            lines = text.splitlines()
            rc = vectors.revcomp_codes(k)
            nclasses = int(((np.arange(4 ** k) <= rc) & (want > 0)).sum())
            assert len(lines) == nclasses
            assert all(len(ln.split(" ")[0]) == k for ln in lines[:50])
            assert np.array_equal(formats.parse_dsk_text(text, k), want)
    # GATB order A<C<T<G: of {ACG.., CGT..} style pairs the printed spelling can differ from lex
    t = formats.dsk_text(np.eye(1, 4 ** 5, mapping.codes_of(["GGGGA"])[0], dtype=np.uint32)[0], 5, "gatb")
    assert t in ("GGGGA 1\n", "TCCCC 1\n") and t == "TCCCC 1\n"      # T < G in GATB's order
    assert formats.dsk_text(np.eye(1, 4 ** 5, mapping.codes_of(["GGGGA"])[0], dtype=np.uint32)[0], 5, "lex") == "GGGGA 1\n"


def test_file_pipeline_batches_and_overlaps_without_a_gpu(tmp_path):
    """pipeline.fastqs_to_images with a stand-in engine (the oracle does the arithmetic): batching
    by size, staging of the next batch on another thread, file naming, stats keys, PNG contents."""
    import threading

    import torch
    from PIL import Image
    from oracle import oracle
    from varkoder_amd import pipeline, synth
    from varkoder_amd.mapping import pixel_lut

    lut = pixel_lut(5, "cgr")
    log = []

    class FakeEngine:
        def stage_files(self, paths, pool=None, slot=0):
            log.append(("stage", slot, [p.name for p in paths], threading.current_thread().name))
            blobs = [p.read_bytes() for p in paths]
            return blobs, slot

        def upload_staged(self, staged, timings=None):
            blobs, slot = staged
            log.append(("upload", slot))
            lens = np.array([len(b) for b in blobs], dtype=np.uint64)
            return blobs, np.zeros(len(blobs), dtype=np.uint64), lens

        def fastq_to_images(self, dev, offs, lens):
            imgs, hists, sts = [], [], []
            for b in dev:
                fwd, _, st = oracle.count_fastq(b, 5)
                imgs.append(oracle.image(oracle.strand_merge(fwd, 5), 5, lut, 1024).reshape(32, 32))
                hists.append(fwd.astype(np.int32))
                sts.append(st)
            return (torch.from_numpy(np.stack(imgs)), torch.from_numpy(np.stack(hists)),
                    torch.from_numpy(np.array(sts, dtype=np.int32)))

        def close(self):
            pass

    files = []
    for i in range(7):
        f = tmp_path / f"s{i}@00000030K.fq"
        f.write_bytes(synth.sample_fastq(i, 200, 150).tobytes())
        files.append(f)
    (tmp_path / "bad@00000001K.fq").write_bytes(b"@r\nACGTACGTAC\n+\nIIIIIIIIII\n@r2\nACGT\n")
    files.append(tmp_path / "bad@00000001K.fq")
    size = files[0].stat().st_size
    stats = pipeline.fastqs_to_images(files, tmp_path / "out", k=5, mapping_code="cgr", batch_bytes=3 * size + 10,
                                      io_threads=2, engine=FakeEngine())
    stages = [e for e in log if e[0] == "stage"]
    assert [len(e[2]) for e in stages] == [3, 3, 2] and [e[1] for e in stages] == [0, 1, 0]
    assert all(e[3] != threading.current_thread().name for e in stages)          # staged off the main thread
    assert [e[1] for e in log if e[0] == "upload"] == [0, 1, 0]
    assert stats["bad@00000001K"]["failed_step"] == "image"
    for i in range(7):
        st = stats[f"s{i}@00000030K"]
        assert "5mer_counting_time" in st and "k5_img_time" in st
        want = oracle.fastq_to_image(synth.sample_fastq(i, 200, 150), 5, lut, 1024)[0].reshape(32, 32)
        im = Image.open(tmp_path / "out" / f"s{i}@00000030K+cgr+k5.png")
        assert np.array_equal(np.array(im), want) and im.info["varkoderMapping"] == "cgr"


def test_basefrequency_sd_equals_reference_cases(tmp_path):
    """image.get_basefrequency_sd against values the reference's own function returned for the same
    fastp-style reports (tests/golden/basesd_cases.json, made by oracle/gen_golden_basesd.py)."""
    import json
    import os

    from varkoder_amd import image
    from varkoder_amd.config import QUAL_THRESH
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "basesd_cases.json")))
    for name, js in gold["reports"].items():
        (tmp_path / name).write_text(json.dumps(js))
    for case in gold["cases"]:
        got = image.get_basefrequency_sd([tmp_path / n for n in case["files"]])
        assert got == pytest.approx(case["base_sd"], rel=1e-12, abs=0), case["files"]
    assert gold["empty_list_returns"] is None and image.get_basefrequency_sd([]) == 0.0   # documented divergence
    # the table the CLI builds from <intermediate>/clean_reads, and the flag it feeds
    cr = tmp_path / "clean_reads"
    cr.mkdir()
    (cr / "good_fastp_unpaired.json").write_text(json.dumps(gold["reports"]["merged_only.json"]))
    (cr / "poor_fastp_paired.json").write_text(json.dumps(gold["reports"]["low_quality.json"]))
    tab = image.base_sd_table(cr, ["good", "poor", "absent"])
    assert tab["good"] < QUAL_THRESH < tab["poor"] and tab["absent"] == 0.0
    assert image.base_sd_table(tmp_path / "nowhere", ["good"]) == {}


def test_stage_files_survives_an_unreadable_file(tmp_path, monkeypatch):
    """ImageEngine.stage_files reads files as they are on disk (gzip files stay compressed: they are
    inflated on the GPU) and stages a vanished or empty one as an empty sample instead of raising out
    of the stager thread (the reference skips such a file and carries on)."""
    import gzip

    import torch
    from varkoder_amd import synth
    from varkoder_amd.engine import ImageEngine
    good = synth.sample_fastq(3, 50, 150).tobytes()
    (tmp_path / "a.fq").write_bytes(good)
    with gzip.open(tmp_path / "b.fq.gz", "wb") as f:
        f.write(good)
    zbytes = (tmp_path / "b.fq.gz").read_bytes()
    (tmp_path / "c.fq.gz").write_bytes(b"\x1f\x8b")           # too short to be a gzip member
    eng = ImageEngine.__new__(ImageEngine)          # no GPU here: only the host half is exercised
    eng.device = 0
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    real_empty = torch.empty
    monkeypatch.setattr(torch, "empty", lambda *a, **k: real_empty(*a, **{x: y for x, y in k.items() if x != "pin_memory"}))
    from varkoder_amd import engine as engine_mod
    paths = [tmp_path / "b.fq.gz", tmp_path / "a.fq", tmp_path / "c.fq.gz", tmp_path / "gone.fq"]
    # plain files are mapped, not read (the GPU copies them out of the page cache: vk_upload_mapped) ...
    monkeypatch.setattr(engine_mod, "USE_MAPPED_UPLOAD", True)
    st = eng.stage_files(paths)
    assert sorted(st["mapped"]) == [1] and bytes(st["mapped"][1][0]) == good and st["offs"][1] == 0
    assert st["mapped"][1][2] is False                 # (no context here: nothing was pinned ahead)
    assert st["disk"].tolist() == [len(zbytes), len(good), 0, 0] and st["lens"].tolist() == [0, len(good), 0, 0]
    st["mapped"][1][1] = None
    st["mapped"][1][0].close()
    # ... or (VARKODER_AMD_MMAP=0; the default for a rank with 16 I/O threads or more) read into the staging buffer like the compressed ones
    monkeypatch.setattr(engine_mod, "USE_MAPPED_UPLOAD", False)
    st = eng.stage_files(paths)
    assert not st["mapped"]
    assert st["is_gz"].tolist() == [True, False, True, False]
    assert st["disk"].tolist() == [len(zbytes), len(good), 0, 0]
    assert st["lens"].tolist() == [0, len(good), 0, 0]              # a gzip file's text length comes from the GPU
    assert st["caps"].tolist() == [len(good), len(good), 0, 0]      # ... its slot from the ISIZE word
    host = st["pinned"].numpy()
    assert st["offs"][1] == 0 and bytes(host[:len(good)]) == good   # plain text first, at its final offset
    assert st["src"][0] >= st["plain_total"] and bytes(host[int(st["src"][0]):int(st["src"][0]) + len(zbytes)]) == zbytes
    assert st["offs"][0] >= st["plain_total"] and st["text_total"] % 16 == 0


def test_plain_route_follows_the_rank_s_io_threads(monkeypatch):
    """engine.plain_route: files come in by DMA from the page cache where a rank has few I/O threads (several ranks
    on one host's cores), through the staging buffer where it has 16 or more; VARKODER_AMD_MMAP forces either."""
    from varkoder_amd import engine as engine_mod
    monkeypatch.setattr(engine_mod, "USE_MAPPED_UPLOAD", None)
    assert [engine_mod.plain_route(t) for t in (1, 2, 8, 15, 16, 64)] == ["mapped"] * 4 + ["staged"] * 2
    monkeypatch.setattr(engine_mod, "USE_MAPPED_UPLOAD", True)
    assert engine_mod.plain_route(64) == "mapped"
    monkeypatch.setattr(engine_mod, "USE_MAPPED_UPLOAD", False)
    assert engine_mod.plain_route(1) == "staged"


def test_bgzf_text_size_walks_block_headers():
    """engine.bgzf_text_size: the text size of a BGZF file without inflating it; None for anything else."""
    import gzip
    import struct
    import zlib
    from varkoder_amd.engine import bgzf_text_size

    def bgzf(data, block):
        out = []
        for i in list(range(0, len(data), block)) + [None]:
            chunk = data[i:i + block] if i is not None else b""
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            raw = co.compress(chunk) + co.flush()
            out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25) +
                       raw + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        return b"".join(out)
    data = b"@r\nACGTACGT\n+\nIIIIIIII\n" * 20000
    blob = bgzf(data, 30000)
    assert gzip.decompress(blob) == data and blob[-4:] == b"\x00\x00\x00\x00"      # the end marker's ISIZE
    assert bgzf_text_size(blob) == len(data)
    assert bgzf_text_size(np.frombuffer(blob, dtype=np.uint8)) == len(data)
    assert bgzf_text_size(gzip.compress(data)) is None          # one ordinary member: its own size word is right
    assert bgzf_text_size(blob + b"junk") is None and bgzf_text_size(blob[:-5]) is None and bgzf_text_size(b"") is None


def test_fastp_shaped_generator_writes_well_formed_fastq():
    """synth.py dist 2 (what bench.py's `realistic` legs and the GPU parity tests count): four-line records,
    sequence and quality of equal length, lengths over the whole range, the oracle accepts it."""
    from oracle import oracle
    from varkoder_amd import synth
    fq = synth.sample_fastq(11, 3000, 150, dist=2)
    hl, ln, off = synth.shaped_layout(11, 3000, 150)
    assert fq.size == off[-1] and (hl >= 40).all() and (hl <= 70).all()
    assert ln.min() == 0 and ln.max() > 250 and 0.55 < (ln == 150).mean() < 0.75 and (ln < 45).mean() > 0.02
    lines = fq.tobytes().split(b"\n")
    assert lines[-1] == b"" and (len(lines) - 1) == 4 * 3000
    for r in range(3000):
        h, s, p, q = lines[4 * r:4 * r + 4]
        assert h.startswith(b"@s00011.%07d " % r) and len(h) + 1 == hl[r]
        assert len(s) == len(q) == ln[r] and p == b"+" and set(s) <= set(b"ACGTN")
    assert any(l[3][:1] == b"@" for l in zip(*[iter(lines[:-1])] * 4))      # a quality line that starts with '@'
    hist, nwin, st = oracle.count_fastq(fq, 7)
    assert st == 0 and nwin == int(hist.sum())


def test_bgzf_table_refuses_blocks_that_claim_more_than_64k_of_text():
    """A BGZF-looking file whose members claim gigabytes of text must not size the text slot: the
    member table is dropped and the file goes the ordinary way, whose slot is clamped against its size on disk."""
    import struct
    import zlib
    from varkoder_amd.engine import bgzf_members

    def block(payload, isize=None):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(payload) + co.flush()
        size = 12 + 6 + len(body) + 8
        hdr = b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, size - 1)
        return hdr + body + struct.pack("<II", zlib.crc32(payload), len(payload) if isize is None else isize)

    good = block(b"@r\nACGT\n+\nIIII\n") + block(b"")
    m = bgzf_members(np.frombuffer(good, dtype=np.uint8))
    assert m is not None and int(m[2].sum()) == 15
    hostile = block(b"@r\nACGT\n+\nIIII\n", isize=0xFFFFFF00) + block(b"")
    assert bgzf_members(np.frombuffer(hostile, dtype=np.uint8)) is None


def test_route_chooser_tries_the_other_route_when_the_first_runs_under_the_link_rate():
    """pipeline.RouteChooser: batches 1-3 on the first route; under 85 % of the link's rate the other route gets its batches,
    and the faster one keeps the rest -- a host whose read() into pinned memory holds the copies back ends up on the
    mapped route by measurement, a fast one never leaves the staged route."""
    import numpy as np
    from varkoder_amd import pipeline

    class FakeEngine:
        route_override = None

        def h2d_link_rate(self):
            return 50e9

    def staged(route):
        return {"disk": np.array([2_000_000_000, 1000], dtype=np.uint64), "is_gz": np.array([False, True]), "plain_route": route}

    def run(rates):
        eng, tm = FakeEngine(), {}
        ch = pipeline.RouteChooser(eng, tm)
        seen = []
        for bi in range(12):
            route = eng.route_override or "staged"
            seen.append(route)
            ch.batch_done(bi, staged(route), 2e9 / rates[route])
        return seen, tm
    seen, tm = run({"staged": 40e9, "mapped": 48e9})
    assert seen[:4] == ["staged"] * 4 and seen[-1] == "mapped" and tm["plain_route_rates_gb_s"]["chosen"] == "mapped"
    seen, tm = run({"staged": 40e9, "mapped": 30e9})
    assert seen[-1] == "staged" and "mapped" in seen and tm["plain_route_rates_gb_s"]["chosen"] == "staged"
    seen, tm = run({"staged": 48e9, "mapped": 10e9})
    assert set(seen) == {"staged"} and tm["plain_route_rates_gb_s"]["chosen"] == "staged" and "mapped" not in tm["plain_route_rates_gb_s"]
