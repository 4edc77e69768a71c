"""NumPy float64 restatement of the reference's image arithmetic -- test infrastructure.

Follows /root/reference/varKoder/commands/image.py:906-919 literally (float64
array, np.quantile linear, np.digitize) so that the integer formulation in
vk_oracle.c can be cross-checked on many more inputs than the golden set.
"""
import numpy as np

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def code_of(kmer):
    c = 0
    for ch in kmer:
        c = c * 4 + "ACGT".index(ch)
    return c


def kmer_of(code, k):
    return "".join("ACGT"[(code >> (2 * (k - 1 - i))) & 3] for i in range(k))


def revcomp_str(s):
    return "".join(COMP[c] for c in reversed(s))


def cgr_xy(k):
    """Float restatement of get_cgr (core/utils.py:185-215) for every code; returns int x, y."""
    corners = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], dtype=float)
    codes = np.arange(4 ** k)
    coords = np.full((codes.size, 2), 0.5)
    for i in range(k):
        b = (codes >> (2 * (k - 1 - i))) & 3
        coords = (coords + corners[b]) / 2
    side = len(np.unique(coords[:, 0]))
    x = (side * (coords[:, 0] - coords[:, 0].min())).astype(int)
    y = (side * (coords[:, 1] - coords[:, 1].min())).astype(int)
    return x, y, side


def image_float(tot, x, y, width, height):
    """image.py:910-919 on per-code class counts `tot` with per-code coordinates x, y."""
    arr = np.zeros(shape=[height, width])
    arr[x, y] = tot.astype(np.float64) + 1
    arr = arr.transpose()
    arr = np.flip(arr, 0)
    bins = np.quantile(arr, np.arange(0, 1, 1 / 256))
    out = np.digitize(arr, bins, right=False) - 1
    return np.uint8(out)


def brute_count(fastq_bytes, k):
    """First-principles counter (pure Python): forward counts u32[4^k], nwindows."""
    lines = fastq_bytes.split(b"\n")
    if lines and lines[-1] == b"":
        lines = lines[:-1]
    fwd = np.zeros(4 ** k, dtype=np.uint32)
    nwin = 0
    for li in range(1, len(lines), 4):
        seq = lines[li].decode("latin-1").upper()
        for i in range(len(seq) - k + 1):
            w = seq[i:i + k]
            if all(ch in "ACGT" for ch in w):
                fwd[code_of(w)] += 1
                nwin += 1
    return fwd, nwin
