"""CPU emulation of the count kernel's wavefront algorithm (tests/emul/lane_emul.cpp, built
around the product's own csrc/vk_lane.h) against the oracle.  No GPU needed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from fastq_cases import edge_cases
from oracle import oracle
from varkoder_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def emul():
    src = os.path.join(HERE, "emul", "lane_emul.cpp")
    so = os.path.join(HERE, "emul", "liblane_emul.so")
    hdr = os.path.join(ROOT, "varkoder_amd", "csrc", "vk_lane.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I",
                               os.path.join(ROOT, "varkoder_amd", "csrc"), src, "-o", so])
    L = C.CDLL(so)
    L.emul_count.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]

    def run(fq, k, parts):
        buf = np.frombuffer(bytes(fq), dtype=np.uint8) if not isinstance(fq, np.ndarray) else fq
        pad = np.zeros(buf.size + 64, dtype=np.uint8)
        pad[:buf.size] = buf
        hist = np.zeros(4 ** k, dtype=np.uint32)
        st = C.c_uint32(0)
        assert L.emul_count(pad.ctypes.data, buf.size, k, parts, hist.ctypes.data, C.byref(st)) == 0
        return hist, st.value

    def run_sampled(fq, k, parts, seed, threshold):
        buf = np.frombuffer(bytes(fq), dtype=np.uint8) if not isinstance(fq, np.ndarray) else fq
        pad = np.zeros(buf.size + 64, dtype=np.uint8)
        pad[:buf.size] = buf
        hist = np.zeros(4 ** k, dtype=np.uint32)
        st = C.c_uint32(0)
        sites = (C.c_uint64 * 2)()
        L.emul_count_sampled.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_uint64, C.c_uint64,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        assert L.emul_count_sampled(pad.ctypes.data, buf.size, k, parts, seed, threshold, hist.ctypes.data,
                                    C.byref(st), sites) == 0
        return hist, st.value, (sites[0], sites[1])
    run.sampled = run_sampled

    def run_dense(fq, k, parts):
        """vk_count_dense_kernel's algorithm (line pass + rounds of listed granules); also returns
        (pieces on the fast path, pieces, rounds, granules counted in rounds)."""
        buf = np.frombuffer(bytes(fq), dtype=np.uint8) if not isinstance(fq, np.ndarray) else fq
        pad = np.zeros(buf.size + 64, dtype=np.uint8)
        pad[:buf.size] = buf
        hist = np.zeros(4 ** k, dtype=np.uint32)
        st = C.c_uint32(0)
        stats = (C.c_uint64 * 6)()
        L.emul_count_dense.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        assert L.emul_count_dense(pad.ctypes.data, buf.size, k, parts, hist.ctypes.data, C.byref(st), stats) == 0
        return hist, st.value, tuple(stats)
    run.dense = run_dense
    return run


@pytest.mark.parametrize("k", (5, 6, 7, 8, 9))
def test_emulated_wave_equals_oracle_on_edge_cases(emul, k):
    for name, fq in edge_cases().items():
        want, nwin, st = oracle.count_fastq(fq, k)
        for parts in (1, 3):
            got, status = emul(fq, k, parts)
            assert status == 0 and st == 0, (name, parts)
            assert np.array_equal(got, want), (name, parts)


@pytest.mark.parametrize("dist", (0, 1))
def test_emulated_wave_equals_oracle_on_synthetic(emul, dist):
    fq = synth.sample_fastq(21, 4000, 150, dist=dist)
    for k in (5, 7, 9):
        want = oracle.count_fastq(fq, k)[0]
        for parts in (1, 2, 7):
            got, status = emul(fq, k, parts)
            assert status == 0
            assert np.array_equal(got, want), (k, parts)


def test_emulated_wave_flags_bad_framing(emul):
    good = synth.sample_fastq(1, 50, 150).tobytes()
    assert emul(good, 7, 1)[1] == 0
    assert emul(good[1:], 7, 1)[1] & 1
    assert emul(good[:-200], 7, 2)[1] & 2
    wrapped = b"".join(b"@r%d\nACGTACGTAC\nGGGTTTAAAC\n+\nIIIIIIIIII\nIIIIIIIIII\n" % i for i in range(40))
    assert emul(wrapped, 7, 1)[1] & 1


def test_emulated_wave_fuzz(emul):
    """Random adversarial (but well-formed) FASTQ: emulated wave algorithm == oracle."""
    from fastq_cases import random_fastq
    rng = np.random.default_rng(2024)
    for trial in range(120):
        fq = random_fastq(rng)
        k = int(rng.integers(5, 10))
        want, nwin, st = oracle.count_fastq(fq, k)
        assert st == 0
        parts = int(rng.integers(1, 4))
        got, status = emul(fq, k, parts)
        assert status == 0, trial
        assert np.array_equal(got, want), (trial, k, parts, len(fq))


def test_emulated_read_subsampling_equals_oracle(emul):
    """The kernels' per-block sampling rule (anchor detection, inheritance across blocks, pieces and
    byte ranges, the position-by-position path for short lines) == the oracle's per-read rule."""
    from fastq_cases import random_fastq, rec
    rng = np.random.default_rng(77)
    blobs = [random_fastq(rng) for _ in range(40)]
    blobs.append(synth.sample_fastq(5, 3000, 150, dist=1).tobytes())
    blobs.append(b"".join(rec(f"r{i}", "ACGTAC") for i in range(3000)))          # > 3 newlines per block
    blobs.append(b"".join(rec(f"q{i}", "ACGT" * 700) for i in range(12)))        # lines longer than a piece
    for j, fq in enumerate(blobs):
        k = 5 + j % 5
        seed = 1000 + j
        for thr in (1 << 31, (1 << 32) // 10, 1 << 32, 0):
            want, nwin, st, wsites = oracle.count_fastq_sampled(fq, k, seed, thr)
            assert st == 0
            for parts in (1, 3):
                got, status, sites = emul.sampled(fq, k, parts, seed, thr)
                assert status == 0 and sites == wsites, (j, thr, parts, sites, wsites)
                assert np.array_equal(got, want), (j, thr, parts)


# ---- the sequence-only heavy stage (vk_count_dense_kernel) ------------------------------------------

@pytest.mark.parametrize("k", (5, 6, 7))
def test_dense_stage_equals_oracle_on_edge_cases(emul, k):
    for name, fq in edge_cases().items():
        want, nwin, st = oracle.count_fastq(fq, k)
        for parts in (1, 3):
            got, status, stats = emul.dense(fq, k, parts)
            assert status == 0 and st == 0, (name, parts)
            assert np.array_equal(got, want), (name, parts, stats)


@pytest.mark.parametrize("dist", (0, 1))
def test_dense_stage_equals_oracle_on_synthetic_and_takes_the_fast_path(emul, dist):
    fq = synth.sample_fastq(21, 40000, 150, dist=dist)   # 12.8 MB: ~190 pieces per wave at parts = 1
    for k, parts in ((5, 1), (7, 1), (7, 3), (6, 7)):
        want = oracle.count_fastq(fq, k)[0]
        got, status, (fast, pieces, rounds, granules, explicit, _noted) = emul.dense(fq, k, parts)
        assert status == 0 and explicit == 0
        assert np.array_equal(got, want), (k, parts)
        # every piece but the first and last of a range goes through the line pass, and the rounds are full
        assert fast >= pieces - 2 * 16 * parts, (fast, pieces)
        assert granules > 60 * rounds
        # sequence bytes are 151 of 320: the heavy stage sees little more than half of the granules
        assert granules * 16 < 0.56 * len(fq), (granules * 16, len(fq))


def test_dense_stage_keeps_fastp_shaped_reads_on_the_fast_path(emul):
    """Reads of every length from 0 to 290 under 40 .. 70 byte headers (synth.py dist 2, what fastp writes for the
    reference's step D): a lane that begins in a quality line and holds its end and a whole header is ordinary
    (seq_span), lanes with four and more newlines are described explicitly -- no piece falls back to the general
    path but the first and last of a range."""
    fq = synth.sample_fastq(33, 30000, 150, dist=2)
    for k, parts in ((7, 1), (5, 2), (6, 5)):
        want = oracle.count_fastq(fq, k)[0]
        got, status, (fast, pieces, rounds, granules, explicit, _noted) = emul.dense(fq, k, parts)
        assert status == 0
        assert np.array_equal(got, want), (k, parts)
        assert fast >= pieces - 2 * 16 * parts, (fast, pieces)
        assert 0 < explicit < 0.02 * 64 * pieces, (explicit, pieces)
    # the same text with every header cut to "@r": records of a few bytes more than their two long lines
    short = b"".join(b"@r\n" + l if i % 4 == 0 else l for i, l in enumerate(x[x.index(b"\n") + 1:] if i % 4 == 0 else x
                     for i, x in enumerate(fq.tobytes().splitlines(keepends=True))))
    want = oracle.count_fastq(short, 7)[0]
    got, status, stats = emul.dense(short, 7, 1)
    assert status == 0 and np.array_equal(got, want), stats


def test_dense_stage_fuzz(emul):
    from fastq_cases import random_fastq, rec
    rng = np.random.default_rng(4242)
    blobs = [random_fastq(rng, nrec=int(rng.integers(1, 400))) for _ in range(150)]
    # reads around the granule and block sizes, headers that end on block borders, long lines
    for n in (13, 14, 15, 16, 17, 30, 31, 32, 33, 45, 47, 48, 49, 62, 63, 64, 65, 66, 127, 128, 129):
        blobs.append(b"".join(rec("r%d" % i, "ACGTTGCA" * (n // 8) + "ACGTTGCA"[: n % 8]) for i in range(700)))
        blobs.append(b"".join(rec("x" * int(rng.integers(1, 70)), "".join(rng.choice(list("ACGTN"), size=n)))
                              for i in range(500)))
    blobs.append(b"".join(rec("q%d" % i, "ACGT" * 3000) for i in range(12)))              # lines longer than a piece
    blobs.append(b"".join(rec("h" * 9000, "ACGTTGCAAC" * 30) for i in range(12)))          # headers longer than a piece
    blobs.append(b"".join(rec("u%d \xc3\xa9" % i, "ACGTTGCAAC" * 15) for i in range(300)))   # UTF-8 in the header
    blobs.append(b"".join(rec("p%d" % i, "A" * 150) for i in range(2000)))                 # low complexity
    for j, fq in enumerate(blobs):
        k = 5 + j % 3
        want, nwin, st = oracle.count_fastq(fq, k)
        assert st == 0
        for parts in (1, 2):
            got, status, stats = emul.dense(fq, k, parts)
            assert status == 0, (j, parts)
            assert np.array_equal(got, want), (j, k, parts, len(fq), stats)
