"""Drop-in executables for the two subprocess contracts of the hot path:

    dsk -nb-cores T -kmer-size K -abundance-min 1 -abundance-min-threshold 1 -max-memory 1000
        -file IN -out-tmp TMP -out OUT                       (commands/image.py:771-790)
    dsk2ascii -c -file IN -nb-cores T -out TMP/dsk.txt -verbose 0     (:875-891, parsed from stdout)

Put `varkoder_amd/shims/bin` first on PATH (`export PATH=$(python -m varkoder_amd.shims):$PATH`)
and the UNMODIFIED reference runs steps D and E of `varKoder image` on the GPU: `dsk` counts with
the HIP library (no CPU fallback; a FASTQ with broken framing exits non-zero, which the reference's
`check=True` turns into `K-MER COUNTING FAIL`), `dsk2ascii` prints the `KMER count` lines the
reference parses.  OUT holds this package's counts container (image.write_counts), not HDF5.
"""
import os
import sys

BIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin")

# options of the two tools that take a value (everything the reference passes, plus the common
# ones of the real tools so that hand-written command lines keep working)
_VALUED = {"-nb-cores", "-kmer-size", "-abundance-min", "-abundance-min-threshold", "-abundance-max",
           "-max-memory", "-max-disk", "-file", "-out-tmp", "-out-dir", "-out", "-verbose", "-solidity-kind",
           "-out-compress", "-storage-type", "-histo-max", "-minimizer-type", "-minimizer-size",
           "-repartition-type"}
_FLAGS = {"-c", "-histo", "-histo2D", "-version", "-help"}


def parse_tool_argv(argv):
    """{option: value} for a dsk / dsk2ascii command line (flags map to True)."""
    opts, i = {}, 0
    while i < len(argv):
        a = argv[i]
        if a in _FLAGS:
            opts[a] = True
            i += 1
        elif a in _VALUED:
            if i + 1 >= len(argv):
                raise SystemExit(f"option {a} needs a value")
            opts[a] = argv[i + 1]
            i += 2
        else:
            raise SystemExit(f"unknown option {a}")
    return opts


def counts_path(out):
    """dsk appends `.h5` when -out has another (or no) extension; the reference always passes
    `<stem>+k<k>.fq.h5` (image.py:752-759)."""
    return out if out.endswith(".h5") else out + ".h5"


def dsk_main(argv=None):
    opts = parse_tool_argv(sys.argv[1:] if argv is None else argv)
    if "-file" not in opts or "-out" not in opts:
        raise SystemExit("dsk: -file and -out are required")
    k = int(opts.get("-kmer-size", 31))
    if not 5 <= k <= 9:
        raise SystemExit(f"dsk (varkoder_amd shim): k-mer size {k} is outside 5..9")
    if int(opts.get("-abundance-min", 1)) != 1:
        raise SystemExit("dsk (varkoder_amd shim): only -abundance-min 1 (no filtering) is supported")
    import numpy as np

    from ..engine import ImageEngine
    from ..image import write_counts
    # the file goes the product's own way: as it is on disk into pinned memory, a .gz inflated on the GPU by
    # vk_inflate_device (one gzip decoder in the product: no host gzip beside it that could disagree with it)
    eng = ImageEngine(k=k, mapping="cgr", device=int(os.environ.get("VARKODER_AMD_DEVICE", "0")))
    try:
        dev, offs, lens = eng.upload_files([opts["-file"]])
        if int(eng.last_upload_status[0]):   # unreadable, or a gzip file the inflate rejected (an EMPTY text is a valid input)
            print(f"dsk (varkoder_amd shim): {opts['-file']} is not a readable FASTQ / gzip file "
                  f"(status {int(eng.last_upload_status[0])})", file=sys.stderr)
            return 1
        h, st = eng.count(dev, offs, lens)
        hist, stw = h.cpu().numpy().view(np.uint32)[0].copy(), int(st.cpu()[0])
        if stw:
            print(f"dsk (varkoder_amd shim): inconsistent FASTQ framing in {opts['-file']} "
                  f"(status bits {stw})", file=sys.stderr)
            return 1
    finally:
        eng.close()
    write_counts(counts_path(opts["-out"]), k, hist)
    return 0


def dsk2ascii_main(argv=None):
    opts = parse_tool_argv(sys.argv[1:] if argv is None else argv)
    if "-file" not in opts:
        raise SystemExit("dsk2ascii: -file is required")
    from ..formats import dsk_text
    from ..image import read_counts
    k, hist = read_counts(opts["-file"])
    text = dsk_text(hist, k, "gatb")
    if opts.get("-c"):
        sys.stdout.write(text)                       # what the reference reads (image.py:893-899)
    elif "-out" in opts:
        with open(opts["-out"], "w") as f:
            f.write(text)
    return 0


if __name__ == "__main__":
    print(BIN_DIR)
