"""Host generator of the synthetic FASTQ workload, bit-identical
to vk_synth_kernel in csrc/vkimg.hip (same counter-based splitmix64 keys).

Record = "@sSSSSS.RRRRRRR\\n" + bases + "\\n+\\n" + "I"*readlen + "\\n" = 2*readlen+20 bytes
(320 at readlen 150).  dist 0 = uniform ACGT; dist 1 = per-sample GC skew plus 1 read in
200 carrying a 20-60 base homopolymer run; N injected at ~1e-3 per base.

dist 2 = reads shaped like what step B of the reference hands to step D (fastp --merge --include_unmerged
--disable_length_filtering, varKoder/commands/image.py:405,426-427,494-495): per read 65 % `readlen` bases,
20 % merged pairs of readlen+1 .. 2*readlen-10, 10 % trimmed reads of 45 .. readlen-1, 5 % of 0 .. 44 bases
(empty reads included); header lines of 40 .. 70 bytes; quality characters '!' .. 'I' ('@' and '+' among
them); bases as dist 0.  Records differ in size: shaped_layout() gives their offsets.
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


_FILL = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ:_/=0123456789", dtype=np.uint8)


def shaped_layout(sample, reads, readlen=150, seed=SEED):
    """dist 2: (header line bytes, bases, record offsets [reads + 1]) of every read of a sample."""
    if not 64 <= readlen <= 1000:
        raise ValueError("dist 2 needs 64 <= readlen <= 1000")
    h = _mix(seed, sample, np.arange(reads, dtype=_U), 0, 5)
    u = h % _U(100)
    v = (h >> _U(8)) & _U(0xFFFFFF)
    ln = np.where(u < 65, _U(readlen), np.where(u < 85, _U(readlen + 1) + v % _U(readlen - 10),
                  np.where(u < 95, _U(45) + v % _U(readlen - 45), v % _U(45)))).astype(np.int64)
    hl = (40 + ((h >> _U(40)) % _U(31))).astype(np.int64)
    off = np.zeros(reads + 1, dtype=np.int64)
    np.cumsum(hl + 2 * ln + 4, out=off[1:])
    return hl, ln, off


def _shaped_fastq(sample, reads, readlen, seed):
    hl, ln, off = shaped_layout(sample, reads, readlen, seed)
    total = int(off[-1])
    o = np.arange(total, dtype=np.int64)
    r = np.searchsorted(off, o, side="right") - 1          # the read every byte belongs to
    p = o - off[r]                                            # its place in the record
    hlr, lnr = hl[r], ln[r]
    out = np.full(total, ord("\n"), dtype=np.uint8)
    # header line: "@sSSSSS.RRRRRRR" + ' ' + filler + newline
    hdr = np.array([f"@s{sample:05d}.{i:07d}".encode() for i in range(reads)], dtype="S15").view(np.uint8).reshape(reads, 15)
    m = p < 15
    out[m] = hdr[r[m], p[m]]
    out[p == 15] = ord(" ")
    m = (p > 15) & (p < hlr - 1)
    out[m] = _FILL[(p[m] + r[m]) % 40]
    out[p == hlr - 1] = ord("\n")     # (a 16-byte header cannot occur: hl >= 40)
    # bases (dist 0's rule, the position inside the read as the key)
    q = p - hlr
    m = (q >= 0) & (q < lnr)
    i, rr = q[m].astype(_U), r[m].astype(_U)
    w, j = i >> _U(4), i & _U(15)
    b = (_mix(seed, sample, rr, w, 1) >> (_U(2) * j)) & _U(3)
    hn = _mix(seed, sample, rr, w, 2)
    isn = (((hn >> _U(8)) & _U(63)) == 0) & ((hn & _U(15)) == j)
    out[m] = np.where(isn, np.uint8(ord("N")), np.frombuffer(b"ACGT", dtype=np.uint8)[b.astype(np.int64)])
    # "\n+\n", quality, "\n"
    q = q - lnr
    out[q == 1] = ord("+")
    q = q - 3
    m = (q >= 0) & (q < lnr)
    i, rr = q[m].astype(_U), r[m].astype(_U)
    out[m] = (33 + ((_mix(seed, sample, rr, i >> _U(3), 6) >> (_U(8) * (i & _U(7)))) & _U(0xFF)) % _U(41)).astype(np.uint8)
    return out


def sample_fastq(sample, reads, readlen=150, seed=SEED, dist=0):
    """uint8 FASTQ text of one sample (reads * (2*readlen+20) bytes for dist 0 and 1)."""
    if dist == 2:
        return _shaped_fastq(sample, reads, readlen, seed)
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
