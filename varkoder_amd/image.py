"""Host-side mirror of the reference's `image` hot-path functions.

Same names, arguments, file naming, return values and error behaviour as
varKoder/commands/image.py count_kmers (:727-806) and make_image (:808-936), but
`dsk` / `dsk2ascii` / pandas / NumPy are replaced by the HIP kernels behind
include/vkimg.h.  PIL is used for the PNG container only.  There is no CPU
fallback: without the HIP library and a GPU these functions raise.
"""
import gzip
import hashlib
import struct
import sys
from collections import OrderedDict
from pathlib import Path

import numpy as np

from .config import BP_KMER_SEP, LABELS_SEP, QUAL_THRESH
from .mapping import lut_from_dataframe

_MAGIC = b"VKH1"  # counts container written at the reference's "<stem>+k<k>.fq.h5" path


def eprint(*args, **kwargs):
    print(*args, file=sys.stderr, **kwargs)  # core/utils.py:39-47


_engines = {}


def _engine(k, lut_key, lut=None, npix=None, device=0):
    """One ImageEngine per (k, mapping table, device) and process."""
    from .engine import ImageEngine
    key = (k, lut_key, device)
    eng = _engines.get(key)
    if eng is None:
        if lut is None:
            eng = ImageEngine(k=k, mapping="cgr", device=device)  # mapping unused by the count stage
        else:
            eng = ImageEngine(k=k, device=device, lut=lut, npix=npix)
        _engines[key] = eng
    return eng


def get_basefrequency_sd(file_list):
    """Quality figure the reference derives from fastp's JSON reports (commands/image.py:45-88): the
    standard deviation of each base's frequency over read cycles 5..39, averaged over A, C, G, T and
    over the report's `merged_and_filtered` / `read1_after_filtering` sections.  As in the reference
    only the FIRST report of the list counts (its return statement sits inside the loop); where the
    reference returns None for an empty list -- and then fails comparing it with the threshold --
    this returns 0.0: no report, no flag."""
    import json
    for f in file_list:
        with open(f, "r") as fh:
            js = json.load(fh)
        sds = []
        for section in ("merged_and_filtered", "read1_after_filtering"):
            try:
                cur = js[section]["content_curves"]
            except KeyError:
                continue
            rows = np.array([x for b, x in cur.items() if b in ("A", "T", "C", "G")])
            sds.append(np.std(rows[:, 5:40], axis=1).mean())
        return float(np.mean(sds)) if sds else float("nan")   # (the reference: np.mean([]) = nan)
    return 0.0


def base_sd_table(clean_reads_dir, samples):
    """{sample: base-frequency sd} from `<sample>_fastp_*.json` in the intermediate clean_reads
    folder (run_clean2img step E, commands/image.py:1094-1097); samples without a report get 0."""
    d = Path(clean_reads_dir)
    if not d.is_dir():
        return {}
    return {s: get_basefrequency_sd(sorted(d.glob(str(s) + "_fastp_*.json"))) for s in samples}


def read_fastq_bytes(infile):
    """Whole FASTQ text of a plain or gzip-compressed file, on the HOST (tests and tools; the product reads
    files through ImageEngine.upload_files, where a .gz is inflated on the GPU)."""
    p = Path(infile)
    with open(p, "rb") as f:
        head = f.read(2)
    if head == b"\x1f\x8b":
        with gzip.open(p, "rb") as f:
            return f.read()
    return p.read_bytes()


def write_counts(path, k, hist):
    with open(path, "wb") as f:
        f.write(_MAGIC + struct.pack("<II", k, hist.size))
        f.write(np.ascontiguousarray(hist, dtype="<u4").tobytes())


def read_counts(path):
    import pandas as pd
    raw = Path(path).read_bytes()
    if len(raw) < 12 or raw[:4] != _MAGIC:
        raise pd.errors.ParserError(f"{path}: not a varkoder_amd k-mer counts file")
    k, n = struct.unpack("<II", raw[4:12])
    if n != 4 ** k or len(raw) != 12 + 4 * n:
        raise pd.errors.ParserError(f"{path}: truncated k-mer counts file")
    return k, np.frombuffer(raw, dtype="<u4", offset=12).astype(np.uint32)


def _stem(path):
    """File name minus ALL suffixes (the reference's naming rule, image.py:753, :840)."""
    p = Path(path)
    return str(p.name.removesuffix("".join(p.suffixes)))


def counts_name(infile, k):
    """`<stem>+k<k>.fq.h5` (image.py:752-759)."""
    return f"{_stem(infile)}{BP_KMER_SEP}k{k}.fq.h5"


def png_name(counts_file, mapping_code):
    """`<sample>@<bp>K+<mapping>+k<k>.png` from a counts file name (image.py:840-849)."""
    base, in_k = _stem(counts_file).split(BP_KMER_SEP)
    return BP_KMER_SEP.join((base, mapping_code, in_k)) + ".png"


def shard_folder(outfolder, outfile, subfolder_levels):
    """md5-hex sharding directories, popped from the END of the digest (image.py:850-854)."""
    outfolder = Path(outfolder)
    if subfolder_levels:
        digest = list(hashlib.md5(outfile.encode("UTF-8")).hexdigest())
        for _ in range(subfolder_levels):
            outfolder = outfolder / digest.pop()
    return outfolder


def write_png(arr, path, labels, base_sd, base_sd_thresh, mapping_code):
    """8-bit "L" PNG, optimize=True, with the four varkoder* text chunks in the reference's order
    (image.py:920-930)."""
    from PIL import Image
    from PIL.PngImagePlugin import PngInfo
    chunks = PngInfo()
    for key, value in (("varkoderKeywords", LABELS_SEP.join(labels)),
                       ("varkoderBaseFreqSd", str(base_sd)),
                       ("varkoderLowQualityFlag", str(base_sd > base_sd_thresh)),
                       ("varkoderMapping", mapping_code)):
        chunks.add_text(key, value)
    Image.fromarray(arr).save(path, optimize=True, pnginfo=chunks)


def _seconds_since(t0):
    import pandas as pd
    return (pd.Timestamp.now() - t0).total_seconds()


def count_kmers(infile, outfolder, threads=1, k=7, overwrite=False, verbose=False):
    """Count k-mers in a FASTQ file (drop-in for commands/image.py:727-806).

    Writes forward-strand counts to `<name minus suffixes>+k<k>.fq.h5` in outfolder (same file
    name rule; the content is this package's own container, only ever read back by make_image).
    Returns OrderedDict {"<k>mer_counting_time": seconds}, or an empty one when the file exists
    and overwrite is False (:761-763).  Raises on failure like the reference's check=True
    subprocess does; `threads` is accepted for signature parity.
    """
    import pandas as pd
    t0 = pd.Timestamp.now()
    Path(outfolder).mkdir(exist_ok=True)
    outpath = Path(outfolder) / counts_name(infile, k)
    if outpath.is_file() and not overwrite:
        eprint("File exists. Skipping kmer counting for file:", str(infile))
        return OrderedDict()

    # the file crosses PCIe as it is on disk; a .gz is inflated on the GPU (dsk reads .gz natively)
    eng = _engine(k, "count")
    dev, offs, lens = eng.upload_files([infile])
    # (the inflate's own status, as the dsk shim reads it: a .gz whose TEXT is empty is a valid input -- dsk exits 0 on
    # it and dsk2ascii dumps nothing, commands/image.py:791-796, :897-899 -- only an unreadable or damaged file fails)
    if int(eng.last_upload_status[0]):
        raise RuntimeError(f"k-mer counting failed for {infile}: not a readable FASTQ / gzip file "
                           f"(status {int(eng.last_upload_status[0])})")
    h, st = eng.count(dev, offs, lens)
    hist, status = h.cpu().numpy().view(np.uint32)[0], int(st.cpu()[0])
    if status:
        raise RuntimeError(f"k-mer counting failed for {infile}: inconsistent FASTQ framing "
                           f"(status bits {status})")
    if verbose:
        eprint(f"vk_count_device k={k} bytes={int(lens[0])} windows={int(hist.sum(dtype=np.uint64))}")
    write_counts(outpath, k, hist)
    return OrderedDict([(f"{k}mer_counting_time", _seconds_since(t0))])


def image_array(hist, kmer_mapping):
    """uint8 [side, side] image of one forward-strand histogram for a reference-style
    mapping DataFrame: the arithmetic of commands/image.py:897-919 on the GPU."""
    k, lut, npix = lut_from_dataframe(kmer_mapping)
    if hist.size != 4 ** k:
        raise IndexError("k-mer counts do not match the k-mer mapping size")
    key = hashlib.sha1(np.ascontiguousarray(lut).tobytes()).hexdigest()
    return _engine(k, key, lut=lut, npix=npix).image_host(hist)


def make_image(infile, outfolder, kmer_mapping, threads=1, overwrite=False, verbose=False, labels=[],
               base_sd=0, base_sd_thresh=QUAL_THRESH, subfolder_levels=0, mapping_code="varKode"):
    """Create an image from k-mer counts (drop-in for commands/image.py:808-936).

    Output name `<sample>@<bp>K+<mapping>+k<k>.png`, optional md5-hex sharding folders,
    skip-if-exists returning an empty OrderedDict (:857-859), PNG + metadata as the reference
    writes them, and {"k<k>_img_time": seconds} (:932-936).  A counts file with no k-mers raises
    pandas' EmptyDataError exactly as the reference's read_csv of an empty dsk2ascii dump does
    (:897-899).
    """
    import pandas as pd
    outfile = png_name(infile, mapping_code)
    folder = shard_folder(outfolder, outfile, subfolder_levels)
    folder.mkdir(exist_ok=True, parents=True)
    if (folder / outfile).is_file() and not overwrite:
        eprint("File exists. Skipping image for file:", str(infile))
        return OrderedDict()

    t0 = pd.Timestamp.now()
    kmer_size = len(kmer_mapping.index[0])  # image.py:862
    k_file, hist = read_counts(infile)
    if k_file != kmer_size:
        raise IndexError(f"{infile}: counted with k={k_file} but the mapping has k={kmer_size}")
    if not hist.any():
        raise pd.errors.EmptyDataError("No columns to parse from file")
    write_png(image_array(hist, kmer_mapping), folder / outfile, labels, base_sd, base_sd_thresh, mapping_code)
    return OrderedDict([(f"k{kmer_size}_img_time", _seconds_since(t0))])
