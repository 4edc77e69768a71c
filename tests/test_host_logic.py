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
