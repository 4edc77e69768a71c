"""Host generator of the synthetic FASTQ workload, bit-identical
to vk_synth_kernel in csrc/vkimg.hip (same counter-based splitmix64 keys).

Record = "@sSSSSS.RRRRRRR\\n" + bases + "\\n+\\n" + "I"*readlen + "\\n" = 2*readlen+20 bytes
(320 at readlen 150).  dist 0 = uniform ACGT; dist 1 = per-sample GC skew plus 1 read in
200 carrying a 20-60 base homopolymer run; N injected at ~1e-3 per base.
"""
import numpy as np

SEED = 20250824
_U = np.uint64


def _mix(seed, s, r, w, stream):
    with np.errstate(over="ignore"):
        z = (_U(seed) + np.asarray(s, dtype=_U) * _U(0x9E3779B97F4A7C15) +
             np.asarray(r, dtype=_U) * _U(0xBF58476D1CE4E5B9) +
             np.asarray(w, dtype=_U) * _U(0x94D049BB133111EB) + _U(stream) * _U(0xD6E8FEB86659FD93))
        z = (z ^ (z >> _U(30))) * _U(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U(27))) * _U(0x94D049BB133111EB)
        return z ^ (z >> _U(31))


def record_bytes(readlen):
    return 2 * readlen + 20


def sample_bases(sample, reads, readlen, seed=SEED, dist=0):
    """uint8[reads, readlen] of ASCII bases for one sample."""
    r = np.arange(reads, dtype=_U)[:, None]
    i = np.arange(readlen, dtype=_U)[None, :]
    w = i >> _U(4)
    j = i & _U(15)
    h1 = _mix(seed, sample, r, w, 1)
    if dist == 0:
        b = (h1 >> (_U(2) * j)) & _U(3)
    else:
        gq = _U(4) + _mix(seed, sample, 0, 0, 3) % _U(7)
        u = (h1 >> (_U(4) * j)) & _U(15)
        at = _U(16) - gq
        a, cc, g = (at + _U(1)) >> _U(1), (gq + _U(1)) >> _U(1), gq >> _U(1)
        b = np.where(u < a, 0, np.where(u < a + cc, 1, np.where(u < a + cc + g, 2, 3))).astype(_U)
        if readlen > 64:
            hr = _mix(seed, sample, r, 0, 4)
            rl = _U(20) + (hr >> _U(16)) % _U(41)
            st = (hr >> _U(32)) % (_U(readlen) - rl)
            inrun = (hr % _U(200) == 0) & (i >= st) & (i < st + rl)
            b = np.where(inrun, (hr >> _U(8)) & _U(3), b)
    hn = _mix(seed, sample, r, w, 2)
    isn = (((hn >> _U(8)) & _U(63)) == 0) & ((hn & _U(15)) == j)
    out = np.frombuffer(b"ACGT", dtype=np.uint8)[b.astype(np.int64)]
    return np.where(isn, np.uint8(ord("N")), out).astype(np.uint8)


def sample_fastq(sample, reads, readlen=150, seed=SEED, dist=0):
    """uint8[reads * (2*readlen+20)] FASTQ text of one sample."""
    rec = record_bytes(readlen)
    buf = np.empty((reads, rec), dtype=np.uint8)
    hdr = np.array([f"@s{sample:05d}.{r:07d}\n".encode() for r in range(reads)], dtype="S16")
    buf[:, :16] = hdr.view(np.uint8).reshape(reads, 16)
    buf[:, 16:16 + readlen] = sample_bases(sample, reads, readlen, seed, dist)
    buf[:, 16 + readlen] = ord("\n")
    buf[:, 17 + readlen] = ord("+")
    buf[:, 18 + readlen] = ord("\n")
    buf[:, 19 + readlen:19 + 2 * readlen] = ord("I")
    buf[:, 19 + 2 * readlen] = ord("\n")
    return buf.ravel()
