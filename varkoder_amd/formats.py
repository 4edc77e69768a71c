"""Intermediate / wire formats next to the hot path.

* `<stem>+k<k>.fq.h5`: this package's counts container (see image.write_counts / read_counts).
* dsk2ascii-style text: one `KMER count` line per canonical class with count >= 1, the format the
  reference parses from `dsk2ascii -c` (commands/image.py:875-899).  The reference's join does not
  depend on which strand spelling is printed; GATB prints the smaller of {s, rc(s)} under ITS 2-bit
  code A0 C1 T2 G3 (from memory -- dsk itself is not available here to confirm), which is what
  `canonical="gatb"` reproduces; `canonical="lex"` prints the lexicographically smaller spelling.
"""
import numpy as np

from .mapping import kmer_strings, revcomp_codes


def _gatb_rank(codes, k):
    """k-mer value under GATB's base order A<C<T<G (our codes are A0 C1 G2 T3)."""
    remap = np.array([0, 1, 3, 2], dtype=np.uint64)
    r = np.zeros(codes.shape, dtype=np.uint64)
    for i in range(k):
        b = (codes >> np.uint32(2 * (k - 1 - i))) & np.uint32(3)
        r = r * np.uint64(4) + remap[b]
    return r


def class_counts(hist, k):
    """(representative codes, counts) of the canonical classes with count >= 1, from a
    forward-strand histogram: count = hist[s] + hist[rc(s)] (hist[s] for a palindrome)."""
    hist = np.asarray(hist, dtype=np.uint64)
    codes = np.arange(4 ** k, dtype=np.uint32)
    rc = revcomp_codes(k)
    tot = np.where(rc == codes, hist, hist + hist[rc])
    return codes, rc, tot


def dsk_text(hist, k, canonical="gatb"):
    """dsk2ascii-style dump of a forward-strand histogram as one string."""
    codes, rc, tot = class_counts(hist, k)
    if canonical == "gatb":
        rep = _gatb_rank(codes, k) <= _gatb_rank(rc, k)
    elif canonical == "lex":
        rep = codes <= rc
    else:
        raise ValueError("canonical must be 'gatb' or 'lex'")
    keep = np.nonzero(rep & (tot > 0))[0]
    names = kmer_strings(k)
    return "".join(f"{names[c]} {int(tot[c])}\n" for c in keep)


def parse_dsk_text(text, k):
    """Inverse of dsk_text for either spelling: per-code class totals u64[4^k] (both spellings of a
    class receive the class count, as after the reference's join with its mapping table)."""
    from .mapping import codes_of
    tot = np.zeros(4 ** k, dtype=np.uint64)
    rc = revcomp_codes(k)
    lines = [ln.split(" ") for ln in text.splitlines() if ln]
    if lines:
        codes = codes_of([a for a, _ in lines])
        vals = np.array([int(b) for _, b in lines], dtype=np.uint64)
        tot[codes] = vals
        tot[rc[codes]] = vals
    return tot
