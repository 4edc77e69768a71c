"""ctypes front-end of oracle/libvkoracle.so (vk_oracle.c) -- test infrastructure.

Each function cites the reference lines it restates in vk_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libvkoracle.so")
    src = os.path.join(_HERE, "vk_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libvkoracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.vko_revcomp.restype = C.c_uint32
        L.vko_revcomp.argtypes = [C.c_uint32, C.c_int]
        L.vko_count_fastq.restype = C.c_int
        L.vko_count_fastq.argtypes = [C.c_void_p, C.c_size_t, C.c_int, u32p, u64p]
        L.vko_strand_merge.restype = None
        L.vko_strand_merge.argtypes = [u32p, C.c_int, u32p]
        L.vko_cgr_lut.restype = None
        L.vko_cgr_lut.argtypes = [C.c_int, u32p]
        L.vko_image.restype = C.c_int
        L.vko_image.argtypes = [u32p, C.c_int, u32p, C.c_uint32, u8p]
        L.vko_fastq_to_image.restype = C.c_int
        L.vko_fastq_to_image.argtypes = [C.c_void_p, C.c_size_t, C.c_int, u32p, C.c_uint32,
                                         u32p, u32p, u8p, u64p]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def count_fastq(data, k):
    """Forward-strand counts u32[4^k] of a FASTQ byte string; returns (fwd, nwindows, status)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    fwd = np.zeros(4 ** k, dtype=np.uint32)
    nwin = C.c_uint64(0)
    st = lib().vko_count_fastq(buf.ctypes.data if buf.size else None, buf.size, k,
                               _p(fwd, C.c_uint32), C.byref(nwin))
    return fwd, nwin.value, st


def count_fastq_sampled(data, k, seed, threshold):
    """count_fastq over the reads taken by the subsampling rule; returns (fwd, nwindows, status,
    (sites_all, sites_taken))."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    fwd = np.zeros(4 ** k, dtype=np.uint32)
    nwin = C.c_uint64(0)
    sites = (C.c_uint64 * 2)()
    L = lib()
    L.vko_count_fastq_sampled.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_uint64,
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    st = L.vko_count_fastq_sampled(buf.ctypes.data if buf.size else None, buf.size, k, seed, threshold,
                                   _p(fwd, C.c_uint32), C.byref(nwin), sites)
    return fwd, nwin.value, st, (sites[0], sites[1])


def strand_merge(fwd, k):
    fwd = np.ascontiguousarray(fwd, dtype=np.uint32)
    tot = np.empty_like(fwd)
    lib().vko_strand_merge(_p(fwd, C.c_uint32), k, _p(tot, C.c_uint32))
    return tot


def cgr_lut(k):
    pix = np.empty(4 ** k, dtype=np.uint32)
    lib().vko_cgr_lut(k, _p(pix, C.c_uint32))
    return pix


def image(tot, k, pix, npix):
    tot = np.ascontiguousarray(tot, dtype=np.uint32)
    pix = np.ascontiguousarray(pix, dtype=np.uint32)
    img = np.empty(npix, dtype=np.uint8)
    st = lib().vko_image(_p(tot, C.c_uint32), k, _p(pix, C.c_uint32), npix, _p(img, C.c_uint8))
    if st != 0:
        raise RuntimeError(f"vko_image status {st}")
    return img


def fastq_to_image(data, k, pix, npix):
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    pix = np.ascontiguousarray(pix, dtype=np.uint32)
    fwd = np.empty(4 ** k, dtype=np.uint32)
    tot = np.empty(4 ** k, dtype=np.uint32)
    img = np.empty(npix, dtype=np.uint8)
    nwin = C.c_uint64(0)
    st = lib().vko_fastq_to_image(buf.ctypes.data, buf.size, k, _p(pix, C.c_uint32), npix,
                                  _p(fwd, C.c_uint32), _p(tot, C.c_uint32), _p(img, C.c_uint8),
                                  C.byref(nwin))
    return img, nwin.value, st
