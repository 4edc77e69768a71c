"""Deterministic, platform-independent test vectors (integer-only splitmix64).

Shared by oracle/gen_golden.py (which records what the reference produced for
them) and the parity tests (which regenerate the same inputs).
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def revcomp_codes(k):
    c = np.arange(4 ** k, dtype=np.uint32)
    r = np.zeros_like(c)
    t = c.copy()
    for _ in range(k):
        r = (r << np.uint32(2)) | (np.uint32(3) - (t & np.uint32(3)))
        t >>= np.uint32(2)
    return r


DISTS = ("heavy", "ties", "sparse", "allequal", "single", "big")


def fwd_hist(k, dist, seed=1):
    """Forward-strand histogram u32[4^k] for one named distribution."""
    n = 4 ** k
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = splitmix64(idx * np.uint64(0x100000001B3) + np.uint64(seed) * np.uint64(0x9E3779B1) +
                       np.uint64(k * 7919 + DISTS.index(dist)))
    if dist == "heavy":
        # geometric exponent (trailing zeros of the high half) times a 10-bit mantissa
        hi = (h >> np.uint64(32)).astype(np.uint64) | np.uint64(1 << 20)
        # exact count-trailing-zeros via the lowest set bit
        with np.errstate(over="ignore"):
            low = hi & (~hi + np.uint64(1))
        e = np.zeros(n, dtype=np.uint64)
        for b in range(21):
            e[low == np.uint64(1 << b)] = b
        v = ((h & np.uint64(0x3FF)) << e) >> np.uint64(4)
        return v.astype(np.uint32)
    if dist == "ties":
        return (h % np.uint64(4)).astype(np.uint32)
    if dist == "sparse":
        v = np.zeros(n, dtype=np.uint32)
        sel = (h % np.uint64(n))[:7].astype(np.int64)
        v[sel] = ((h[7:14] % np.uint64(100000)) + np.uint64(1)).astype(np.uint32)
        return v
    if dist == "allequal":
        return np.full(n, 7, dtype=np.uint32)
    if dist == "single":
        v = np.zeros(n, dtype=np.uint32)
        v[int(h[0] % np.uint64(n))] = 12345
        return v
    if dist == "big":
        v = (h % np.uint64(1000)).astype(np.uint32)
        sel = (h[:64] % np.uint64(n)).astype(np.int64)
        v[sel] = np.uint32(2 ** 30) - (h[64:128] % np.uint64(5000)).astype(np.uint32)
        return v
    raise ValueError(dist)


def class_totals(fwd, k):
    """tot[c] = windows whose canonical class is {c, rc(c)} (palindromes once)."""
    rc = revcomp_codes(k)
    fwd = fwd.astype(np.uint64)
    tot = np.where(rc == np.arange(4 ** k), fwd, fwd + fwd[rc])
    assert tot.max() < 2 ** 32 - 1
    return tot.astype(np.uint32)


def random_image(side, seed):
    """Asymmetric pseudo-random uint8 image (used to pin remap's duplicate-resolution order)."""
    idx = np.arange(side * side, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = splitmix64(idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed))
    return (h & np.uint64(0xFF)).astype(np.uint8).reshape(side, side)
