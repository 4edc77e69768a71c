"""The argv-level drop-in (SURVEY.md 8b "argv contracts"): `dsk` / `dsk2ascii` stand-ins that take
the reference's own command lines (golden `tool_argv`, captured from commands/image.py:771-790 and
:875-891 by oracle/gen_golden.py)."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import fastq_cases
import vectors
from oracle import oracle
from varkoder_amd import formats, image, shims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tool_argv():
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        return json.load(f)["tool_argv"]


def _fill(call, **paths):
    return [a.format(**paths) if "{" in a else a for a in call]


def test_reference_command_lines_parse(tool_argv):
    dsk, d2a = tool_argv["calls"]
    assert dsk[0] == "dsk" and d2a[0] == "dsk2ascii"
    o = shims.parse_tool_argv(_fill(dsk, IN="in.fq.gz", TMP="/tmp/x", COUNTS="c/S+k7.fq.h5")[1:])
    assert o["-file"] == "in.fq.gz" and o["-out"] == "c/S+k7.fq.h5" and o["-kmer-size"] == str(tool_argv["k"])
    assert o["-abundance-min"] == "1" and o["-abundance-min-threshold"] == "1" and o["-max-memory"] == "1000"
    assert o["-nb-cores"] == str(tool_argv["threads_count"]) and o["-out-tmp"] == "/tmp/x"
    o = shims.parse_tool_argv(_fill(d2a, TMP="/tmp/x", COUNTS="c/S+k7.fq.h5")[1:])
    assert o["-c"] is True and o["-file"] == "c/S+k7.fq.h5" and o["-out"] == "/tmp/x/dsk.txt"
    with pytest.raises(SystemExit):
        shims.parse_tool_argv(["-no-such-option"])


def test_bench_times_the_reference_command_lines(tool_argv):
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dsk, d2a = tool_argv["calls"]
    assert bench.dsk_argv(tool_argv["threads_count"], tool_argv["k"], "I", "T", "C") == _fill(dsk, IN="I", TMP="T", COUNTS="C")
    assert bench.dsk2ascii_argv(tool_argv["threads_image"], "C", "T") == _fill(d2a, TMP="T", COUNTS="C")


def test_shims_are_on_path_as_the_tool_names():
    for name in ("dsk", "dsk2ascii"):
        p = os.path.join(shims.BIN_DIR, name)
        assert os.access(p, os.X_OK), p
    out = subprocess.run([sys.executable, "-m", "varkoder_amd.shims"], capture_output=True, text=True, cwd=ROOT)
    assert out.stdout.strip() == shims.BIN_DIR


@pytest.mark.parametrize("k", [5, 8])
def test_dsk2ascii_shim_prints_what_the_reference_parses(tmp_path, tool_argv, k):
    hist = vectors.fwd_hist(k, "sparse" if k == 8 else "heavy")
    counts = tmp_path / f"S@00010000K+k{k}.fq.h5"
    image.write_counts(counts, k, hist)
    argv = _fill(tool_argv["calls"][1], TMP=str(tmp_path), COUNTS=str(counts))
    out = subprocess.run([os.path.join(shims.BIN_DIR, argv[0])] + argv[1:], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert all(len(a) == k and set(a) <= set("ACGT") and int(b) >= 1 for a, b in (ln.split(" ") for ln in lines))
    _, _, want = formats.class_counts(hist, k)
    assert np.array_equal(formats.parse_dsk_text(out.stdout, k), want)
    assert not (tmp_path / "dsk.txt").exists()      # like the real tool with -c: stdout only


def test_dsk2ascii_shim_rejects_a_foreign_file(tmp_path, tool_argv):
    bad = tmp_path / "x+k7.fq.h5"
    bad.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    argv = _fill(tool_argv["calls"][1], TMP=str(tmp_path), COUNTS=str(bad))
    out = subprocess.run([os.path.join(shims.BIN_DIR, argv[0])] + argv[1:], capture_output=True, text=True)
    assert out.returncode != 0 and out.stdout == ""


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["qual_starts_at_plus", "mixed_case_iupac", "no_final_newline"])
def test_dsk_shim_counts_on_the_gpu(tmp_path, tool_argv, case):
    data = fastq_cases.edge_cases()[case]
    fq = tmp_path / "S1@00000001K.fq"
    fq.write_bytes(data)
    k = tool_argv["k"]
    counts = tmp_path / f"S1@00000001K+k{k}.fq.h5"
    dsk, d2a = tool_argv["calls"]
    a = _fill(dsk, IN=str(fq), TMP=str(tmp_path), COUNTS=str(counts))
    r = subprocess.run([os.path.join(shims.BIN_DIR, a[0])] + a[1:], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    a = _fill(d2a, TMP=str(tmp_path), COUNTS=str(counts))
    r = subprocess.run([os.path.join(shims.BIN_DIR, a[0])] + a[1:], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want_fwd = oracle.count_fastq(data, k)[0]
    _, _, want = formats.class_counts(want_fwd, k)
    assert np.array_equal(formats.parse_dsk_text(r.stdout, k), want)


@pytest.mark.gpu
def test_dsk_shim_fails_like_check_true_on_broken_framing(tmp_path, tool_argv):
    fq = tmp_path / "bad@00000001K.fq"
    fq.write_bytes(b"@r\nACGTACGTACGT\n+\nIIIIIIIIIIII\n@r2\nACGTACGTAC\n")     # truncated record
    counts = tmp_path / "bad@00000001K+k7.fq.h5"
    a = _fill(tool_argv["calls"][0], IN=str(fq), TMP=str(tmp_path), COUNTS=str(counts))
    r = subprocess.run([os.path.join(shims.BIN_DIR, a[0])] + a[1:], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not counts.exists()
    assert "framing" in r.stderr


@pytest.mark.gpu
def test_dsk_shim_takes_an_empty_text_gz_and_rejects_a_damaged_one(tmp_path, tool_argv):
    """The shim reads through the engine's mapped route (one I/O thread): a valid .gz whose text is empty is a
    sample without k-mers (dsk would write an empty table), not an unreadable file; a damaged .gz is an error."""
    import gzip
    k = tool_argv["k"]
    dsk, d2a = tool_argv["calls"]
    empty = tmp_path / "E@00000001K.fq.gz"
    empty.write_bytes(gzip.compress(b""))
    counts = tmp_path / f"E@00000001K+k{k}.fq.h5"
    a = _fill(dsk, IN=str(empty), TMP=str(tmp_path), COUNTS=str(counts))
    r = subprocess.run([os.path.join(shims.BIN_DIR, a[0])] + a[1:], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and counts.exists(), r.stderr
    a = _fill(d2a, TMP=str(tmp_path), COUNTS=str(counts))
    r = subprocess.run([os.path.join(shims.BIN_DIR, a[0])] + a[1:], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout == ""
    # a real .gz through the same (mapped) route
    data = fastq_cases.edge_cases()["mixed_case_iupac"]
    good = tmp_path / "G@00000001K.fq.gz"
    good.write_bytes(gzip.compress(data))
    counts = tmp_path / f"G@00000001K+k{k}.fq.h5"
    a = _fill(dsk, IN=str(good), TMP=str(tmp_path), COUNTS=str(counts))
    r = subprocess.run([os.path.join(shims.BIN_DIR, a[0])] + a[1:], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    a = _fill(d2a, TMP=str(tmp_path), COUNTS=str(counts))
    r = subprocess.run([os.path.join(shims.BIN_DIR, a[0])] + a[1:], capture_output=True, text=True, timeout=300)
    _, _, want = formats.class_counts(oracle.count_fastq(data, k)[0], k)
    assert np.array_equal(formats.parse_dsk_text(r.stdout, k), want)
    # damaged: the shim fails like `check=True` would see dsk fail
    bad = tmp_path / "B@00000001K.fq.gz"
    blob = bytearray(gzip.compress(data * 20))
    blob[len(blob) // 2] ^= 0xFF
    bad.write_bytes(bytes(blob))
    counts = tmp_path / f"B@00000001K+k{k}.fq.h5"
    a = _fill(dsk, IN=str(bad), TMP=str(tmp_path), COUNTS=str(counts))
    r = subprocess.run([os.path.join(shims.BIN_DIR, a[0])] + a[1:], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not counts.exists()
