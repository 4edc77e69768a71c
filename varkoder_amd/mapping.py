"""k-mer -> pixel mappings (reference: varKoder/core/utils.py:152-217 and
varKoder/kmer_mapping/<k>mer_mapping.parquet).

Two views of the same thing:
  * pixel_lut(k, method): u32[4^k], image pixel index (row-major, row 0 on top) of
    every k-mer code -- what the HIP kernels consume (include/vkimg.h);
  * get_kmer_mapping(k, method): the reference's DataFrame (index = k-mer string,
    int columns x, y), for callers that pass it to make_image() like the reference
    does (commands/image.py:1229, 1099-1110).
Code convention: A0 C1 G2 T3, first base most significant.
"""
import functools
import os

import numpy as np

from .config import KMER_MAX, KMER_MIN

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "varkode_lut.npz")
# image side of the shipped varKode tables (= max(x)+1 of the parquet tables)
VARKODE_SIDE = {5: 23, 6: 46, 7: 91, 8: 182, 9: 363}


def _check_k(k):
    if k not in range(KMER_MIN, KMER_MAX + 1):
        raise ValueError("kmer size must be between 5 and 9")  # commands/image.py:1209-1210


def side(k, method):
    _check_k(k)
    if method == "cgr":
        return 2 ** k
    if method == "varKode":
        return VARKODE_SIDE[k]
    raise Exception('method must be "varKode" or "cgr"')  # core/utils.py:169


def cgr_xy(k):
    """Closed form of get_cgr (core/utils.py:185-215): with b_i the code of base i,
    x = sum ((b_i>>1)&1) 2^i, y = sum (((b_i>>1)^b_i)&1) 2^i  (first base = bit 0)."""
    codes = np.arange(4 ** k, dtype=np.uint32)
    x = np.zeros_like(codes)
    y = np.zeros_like(codes)
    for i in range(k):
        b = (codes >> np.uint32(2 * (k - 1 - i))) & np.uint32(3)
        x |= ((b >> 1) & 1) << np.uint32(i)
        y |= (((b >> 1) ^ b) & 1) << np.uint32(i)
    return x, y


@functools.lru_cache(maxsize=None)
def pixel_lut(k, method):
    """u32[4^k]: pixel index (side-1-y)*side + x of every code (image.py:906-913)."""
    n = side(k, method)
    if method == "cgr":
        x, y = cgr_xy(k)
        lut = ((n - 1 - y.astype(np.int64)) * n + x).astype(np.uint32)
    else:
        with np.load(_DATA) as z:
            lut = z[f"k{k}"].astype(np.uint32)
    lut.setflags(write=False)
    return lut


def revcomp_codes(k):
    c = np.arange(4 ** k, dtype=np.uint32)
    r = np.zeros_like(c)
    t = c.copy()
    for _ in range(k):
        r = (r << np.uint32(2)) | (np.uint32(3) - (t & np.uint32(3)))
        t >>= np.uint32(2)
    return r


def kmer_strings(k):
    codes = np.arange(4 ** k, dtype=np.uint32)
    chars = np.empty((4 ** k, k), dtype="S1")
    alphabet = np.array([b"A", b"C", b"G", b"T"], dtype="S1")
    for i in range(k):
        chars[:, i] = alphabet[(codes >> np.uint32(2 * (k - 1 - i))) & np.uint32(3)]
    return chars.view(f"S{k}").ravel().astype(str)


def codes_of(kmers):
    """Codes of an iterable of equal-length ACGT strings."""
    kmers = list(kmers)
    k = len(kmers[0])
    arr = np.frombuffer("".join(kmers).encode("ascii"), dtype=np.uint8).reshape(-1, k)
    b = np.full(arr.shape, 255, dtype=np.uint32)
    for ch, v in zip(b"ACGT", range(4)):
        b[arr == ch] = v
    if (b == 255).any():
        raise ValueError("k-mer mapping index must contain only A, C, G, T")
    code = np.zeros(arr.shape[0], dtype=np.uint32)
    for i in range(k):
        code = code * np.uint32(4) + b[:, i]
    return code


def get_cgr(kmer_size):
    """DataFrame like the reference's get_cgr: 2*4^k rows, every k-mer at its own
    coordinates and every reverse-complement spelling at the original's."""
    import pandas as pd
    _check_k(kmer_size)
    x, y = cgr_xy(kmer_size)
    names = kmer_strings(kmer_size)
    rc = revcomp_codes(kmer_size)
    df = pd.concat([pd.DataFrame({"x": x.astype(int), "y": y.astype(int)}, index=names),
                    pd.DataFrame({"x": x.astype(int), "y": y.astype(int)}, index=names[rc])])
    df.attrs["vk_method"] = "cgr"
    df.attrs["vk_k"] = kmer_size
    return df


def get_kmer_mapping(kmer_size=7, method="varKode"):
    """Drop-in for core/utils.py:152-171 (same defaults, same exception text)."""
    import pandas as pd
    if method == "varKode":
        _check_k(kmer_size)
        n = side(kmer_size, method)
        lut = pixel_lut(kmer_size, method).astype(np.int64)
        df = pd.DataFrame({"x": (lut % n).astype(np.int32), "y": (n - 1 - lut // n).astype(np.int32)},
                          index=pd.Index(kmer_strings(kmer_size), name="kmer"))
    elif method == "cgr":
        df = get_cgr(kmer_size)
    else:
        raise Exception('method must be "varKode" or "cgr"')
    df.attrs["vk_method"] = method
    df.attrs["vk_k"] = kmer_size
    return df


def lut_from_dataframe(kmer_mapping):
    """(k, lut, npix) from a reference-style mapping DataFrame.  Tables made by
    get_kmer_mapping are recognised by their attrs; any other table must give every
    k-mer s the pixel set {P(s), P(rc s)} for some per-k-mer pixel P (true of both
    reference mappings), otherwise the count+1 scatter is not expressible per code."""
    k = len(kmer_mapping.index[0])
    _check_k(k)
    meth = kmer_mapping.attrs.get("vk_method")
    if meth in ("cgr", "varKode") and kmer_mapping.attrs.get("vk_k") == k:
        n = side(k, meth)
        return k, pixel_lut(k, meth), n * n
    width = int(kmer_mapping["x"].max()) + 1
    height = int(kmer_mapping["y"].max()) + 1
    if width != height:
        raise ValueError("only square k-mer mappings are supported")
    codes = codes_of(kmer_mapping.index)
    pix = ((height - 1 - kmer_mapping["y"].to_numpy().astype(np.int64)) * width +
           kmer_mapping["x"].to_numpy().astype(np.int64)).astype(np.uint32)
    lut = np.full(4 ** k, 0xFFFFFFFF, dtype=np.uint32)
    # first occurrence wins (own coordinates in the reference's cgr table)
    order = np.arange(len(codes))[::-1]
    lut[codes[order]] = pix[order]
    if (lut == 0xFFFFFFFF).any():
        raise ValueError("k-mer mapping does not cover every k-mer")
    rc = revcomp_codes(k)
    ok = (pix == lut[codes]) | (pix == lut[rc[codes]])
    if not ok.all():
        raise ValueError("unsupported k-mer mapping: a k-mer maps outside {P(s), P(rc s)}")
    return k, lut, width * height
