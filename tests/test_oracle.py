"""CPU suite: the oracle (oracle/vk_oracle.c) against the reference's golden vectors,
against a literal float64 restatement, and against first-principles counting."""
import hashlib

import numpy as np
import pytest

import vectors
from fastq_cases import edge_cases, rec
from oracle import np_oracle, oracle
from varkoder_amd import synth
from varkoder_amd.mapping import pixel_lut, side

KS = (5, 6, 7, 8, 9)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---- image stage: pinned on the reference's own make_image output -----------------------

@pytest.mark.parametrize("k", KS)
@pytest.mark.parametrize("mapping", ("cgr", "varKode"))
def test_oracle_image_equals_reference_golden(manifest, golden_small, k, mapping):
    lut, n = pixel_lut(k, mapping), side(k, mapping)
    for dist in vectors.DISTS:
        fwd = vectors.fwd_hist(k, dist)
        tot = oracle.strand_merge(fwd, k)
        assert np.array_equal(tot, vectors.class_totals(fwd, k))
        img = oracle.image(tot, k, lut, n * n).reshape(n, n)
        key = f"k{k}_{mapping}_{dist}"
        assert list(img.shape) == manifest["image_cases"][key]["shape"]
        if key in golden_small:
            assert np.array_equal(img, golden_small[key]), key
        assert sha(img) == manifest["image_cases"][key]["sha256"], key


@pytest.mark.parametrize("k", KS)
def test_luts_match_reference(manifest, k):
    assert sha(oracle.cgr_lut(k)) == manifest["cgr_lut"][str(k)]["sha256"]
    assert sha(pixel_lut(k, "cgr")) == manifest["cgr_lut"][str(k)]["sha256"]
    assert sha(pixel_lut(k, "varKode")) == manifest["varkode_lut"][str(k)]["sha256"]
    assert side(k, "varKode") == manifest["varkode_lut"][str(k)]["side"]
    # s and rc(s) share a varKode pixel; #pixels = #canonical classes
    lut = pixel_lut(k, "varKode")
    rc = vectors.revcomp_codes(k)
    assert np.array_equal(lut, lut[rc])
    assert len(np.unique(lut)) == manifest["varkode_lut"][str(k)]["distinct_pixels"]


@pytest.mark.parametrize("k", (5, 6, 7))
def test_integer_binning_equals_float64_numpy(k):
    """np.quantile(linear) + np.digitize in float64 (image.py:916-919) == the integer form."""
    rng = np.random.default_rng(k)
    x, y, n = np_oracle.cgr_xy(k)
    lut = pixel_lut(k, "cgr")
    assert np.array_equal(((n - 1 - y) * n + x).astype(np.uint32), lut)
    for trial in range(6):
        fwd = rng.negative_binomial(1 + trial, 0.02 + 0.1 * trial, size=4 ** k).astype(np.uint32)
        if trial == 4:
            fwd[rng.random(4 ** k) < 0.97] = 0
        tot = oracle.strand_merge(fwd, k)
        want = np_oracle.image_float(tot, x, y, n, n)
        got = oracle.image(tot, k, lut, n * n).reshape(n, n)
        assert np.array_equal(got, want), trial


# ---- counting stage: first principles (dsk itself cannot run here: parity unpinned) ------

def test_known_answer_counts():
    k = 3
    fq = rec("r", "ACGTN")  # windows ACG, CGT; GTN and TN* contain N
    fwd, nwin, st = oracle.count_fastq(fq, k)
    assert st == 0 and nwin == 2
    assert fwd[np_oracle.code_of("ACG")] == 1 and fwd[np_oracle.code_of("CGT")] == 1
    # strand merge: ACG and CGT are reverse complements -> one class with count 2
    tot = oracle.strand_merge(fwd, k)
    assert tot[np_oracle.code_of("ACG")] == 2 and tot[np_oracle.code_of("CGT")] == 2
    # palindrome at even k counted once
    fwd, nwin, st = oracle.count_fastq(rec("p", "ACGT"), 4)
    assert nwin == 1 and oracle.strand_merge(fwd, 4)[np_oracle.code_of("ACGT")] == 1
    # read shorter than k contributes nothing; lower case counts like upper case
    assert oracle.count_fastq(rec("s", "ACGTAC"), 7)[1] == 0
    up = oracle.count_fastq(rec("s", "ACGTACGTAC"), 5)[0]
    lo = oracle.count_fastq(rec("s", "acgtacgtac"), 5)[0]
    assert np.array_equal(up, lo)


@pytest.mark.parametrize("k", (5, 7, 9))
def test_oracle_count_equals_bruteforce_on_edge_cases(k):
    for name, fq in edge_cases().items():
        if len(fq) > 400000:
            continue
        want, nw = np_oracle.brute_count(fq.replace(b"\r", b"?"), k)
        got, nwin, st = oracle.count_fastq(fq, k)
        assert st == 0, name
        assert nwin == nw and np.array_equal(got, want), name


def test_window_count_property():
    """sum(counts) = sum(max(0, len-k+1)) - windows touching a non-ACGT byte."""
    fq = synth.sample_fastq(11, 500, 150, dist=1)
    bases = synth.sample_bases(11, 500, 150, dist=1)
    for k in (5, 7, 9):
        isn = (bases == ord("N")).astype(np.int64)
        cs = np.concatenate([np.zeros((500, 1), np.int64), np.cumsum(isn, axis=1)], axis=1)
        clean = (cs[:, k:] - cs[:, :-k]) == 0
        fwd, nwin, st = oracle.count_fastq(fq, k)
        assert st == 0 and nwin == int(clean.sum()) == int(fwd.sum(dtype=np.uint64))


def test_format_errors_flagged():
    good = rec("a", "ACGTACGTACGT")
    assert oracle.count_fastq(good, 5)[2] == 0
    assert oracle.count_fastq(good[1:], 5)[2] != 0           # no '@'
    assert oracle.count_fastq(good[:-8], 5)[2] == 0          # quality line cut short is still 4 lines
    assert oracle.count_fastq(good[:17], 5)[2] != 0          # truncated before the quality line


def test_host_generator_is_stable():
    fq = synth.sample_fastq(3, 4, 150)
    assert fq.size == 1280 and bytes(fq[:16]) == b"@s00003.0000000\n"
    assert sha(fq) == sha(synth.sample_fastq(3, 4, 150))
    # pinned content hash: the device generator (vk_synth_kernel) must reproduce these bytes
    assert sha(synth.sample_fastq(0, 100, 150, dist=0))[:16] == SYNTH_SHA0
    assert sha(synth.sample_fastq(7, 100, 150, dist=1))[:16] == SYNTH_SHA1


SYNTH_SHA0 = "d3e00ad16ca1838a"
SYNTH_SHA1 = "c75101c30b4a16db"


@pytest.mark.parametrize("k", [5, 6, 7, 8, 9])
def test_edge_case_dsk_text_goldens_are_current(k):
    """tests/golden/dsk_text_k<k>/*.txt (oracle/gen_golden_dsktext.py): the dsk2ascii-style dump of every
    edge case as this build counts it, for every k the package claims -- kept so that a machine with GATB
    dsk 2.3.3 can close the 'counting parity unpinned' gap with one diff per case."""
    import os

    from fastq_cases import dsk_kit_cases
    from varkoder_amd import formats
    gold = os.path.join(os.path.dirname(__file__), "golden", f"dsk_text_k{k}")
    cases = dsk_kit_cases(k)
    assert {f[:-4] for f in os.listdir(gold) if f.endswith(".txt")} == set(cases)
    for name, fq in cases.items():
        hist, nwin, st = oracle.count_fastq(fq, k)
        assert st == 0
        text = formats.dsk_text(hist, k, "gatb")
        with open(os.path.join(gold, name + ".txt")) as f:
            assert f.read() == text, name
        # the dump is canonical: every class once, under its GATB spelling, counts summing to the windows
        assert sum(int(line.split()[1]) for line in text.splitlines()) == nwin
        p = os.path.join(gold, name + ".fq")
        if os.path.exists(p):
            with open(p, "rb") as f:
                assert f.read() == fq, name
    if k in (6, 8):   # even k: the palindrome case really holds palindromes, each dumped once with its full count
        hist = oracle.count_fastq(cases["palindromes_even_k"], k)[0]
        from varkoder_amd.mapping import revcomp_codes
        nz = np.nonzero(hist)[0]
        pal = nz[revcomp_codes(k)[nz] == nz]
        assert len(pal), "no palindromic k-mer in the palindrome case"
