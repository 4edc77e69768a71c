"""Hand-written FASTQ edge cases shared by the oracle tests and the GPU parity tests."""
import numpy as np


def rec(name, seq, qual=None, plus="+"):
    q = qual if qual is not None else "I" * len(seq)
    return f"@{name}\n{seq}\n{plus}\n{q}\n".encode()


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(list(alphabet), size=n))


def edge_cases():
    rng = np.random.default_rng(7)
    cases = {}
    cases["empty"] = b""
    cases["one_read"] = rec("r", "ACGTACGTACGTAAACCCGGGTTT")
    cases["n_in_middle"] = rec("r", "ACGTACGTANACGTACGTACGT")
    cases["len_k_minus_1"] = rec("r", "ACGTAC")          # k=7: nothing
    cases["len_k"] = rec("r", "ACGTACG")
    cases["lowercase"] = rec("r", "acgtacgtacgtnacgtacgtacg")
    cases["mixed_case_iupac"] = rec("r", "ACGTRYKMacgtACGTACGTSWBDHVNACGTACGTAC")
    cases["crlf"] = b"@r\r\nACGTACGTACGTACGT\r\n+\r\nIIIIIIIIIIIIIIII\r\n"
    cases["no_final_newline"] = rec("a", "ACGTACGTACGTTTGA") + b"@b\nGGGGGGGGGGCCCCCCCCCC\n+\nIIIIIIIIIIIIIIIIIIII"
    cases["qual_starts_at_plus"] = (rec("a", "ACGTACGTACGT", qual="@+@+@+@+@+@+") +
                                    rec("b", "TTTTACGTAAAC", qual="+@+@+@+@+@+@", plus="+b") +
                                    rec("c", "GATTACAGATTACA", qual="@IIIIIIIIIIIII"))
    cases["palindromes_even_k"] = rec("p", "ACGTACGTACGTACGTAATTAATTGGCCGGCCGCGCGCGCATATATAT")
    cases["poly_a"] = rec("p", "A" * 300)
    cases["empty_read"] = rec("e", "") + rec("f", "ACGTACGTACGTACG")
    cases["long_header"] = rec("h" * 700 + " ACGTACGTACGTACGT", "GATTACAGATTACAGATTACA")
    cases["long_read"] = rec("long", rand_seq(rng, 20000))
    cases["many_short"] = b"".join(rec(f"s{i}", rand_seq(rng, int(rng.integers(0, 24)), "ACGTN")) for i in range(3000))
    cases["ragged"] = b"".join(rec(f"r{i} extra/1", rand_seq(rng, int(rng.integers(30, 260)), "ACGTACGTACGTN"))
                               for i in range(4000))
    cases["exact_64_multiple"] = b"".join(rec("%06d" % i, rand_seq(rng, 24)) for i in range(1024))  # 64 B records
    return cases


LARGE_CASES = ("long_read", "many_short", "ragged", "exact_64_multiple")


def dsk_kit_cases(k):
    """The cases of the dsk closing kit at k (oracle/gen_golden_dsktext.py): all of edge_cases() at k = 7; at the
    other k the hand-written ones, with `len_k` / `len_k_minus_1` re-made for that k."""
    cases = edge_cases()
    if k != 7:
        for name in LARGE_CASES:
            del cases[name]
        cases["len_k"] = rec("r", "ACGTACGTAC"[:k])
        cases["len_k_minus_1"] = rec("r", "ACGTACGTAC"[:k - 1])
    return cases


def random_fastq(rng, nrec=None):
    """A structurally valid 4-line FASTQ with adversarial content: zero-length and very long reads,
    IUPAC / lower-case / punctuation in sequences, '@' and '+' leading quality lines, long or empty
    header tails, optional CRLF, optional missing final newline."""
    nrec = int(rng.integers(1, 60)) if nrec is None else nrec
    alph = [list("ACGT"), list("ACGTN"), list("ACGTacgtNnRYKM.-*"), list("AC")]
    out = []
    crlf = rng.random() < 0.15
    eol = b"\r\n" if crlf else b"\n"
    for i in range(nrec):
        a = alph[int(rng.integers(0, len(alph)))]
        mode = rng.random()
        n = 0 if mode < 0.05 else int(rng.integers(1, 12)) if mode < 0.2 else int(rng.integers(12, 400)) \
            if mode < 0.95 else int(rng.integers(400, 6000))
        seq = "".join(rng.choice(a, size=n)) if n else ""
        q0 = rng.choice(list("@+I5#"))
        qual = (q0 + "".join(rng.choice(list("!#5ACGT@+IJ~"), size=max(n - 1, 0))))[:n]
        hdr = "@" + "".join(rng.choice(list("abcXYZ012:/ _ACGT@+"), size=int(rng.integers(0, 90))))
        plus = "+" + (hdr[1:] if rng.random() < 0.2 else "")
        out.append(hdr.encode() + eol + seq.encode() + eol + plus.encode() + eol + qual.encode() + eol)
    data = b"".join(out)
    if rng.random() < 0.3 and not crlf and n > 0:
        data = data[:-1]  # no final newline (only meaningful when the last quality line is not empty)
    return data
