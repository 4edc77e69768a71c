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

from . import _capi
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


def read_fastq_bytes(infile):
    """Whole FASTQ text of a plain or gzip-compressed file (dsk reads .gz natively)."""
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


def count_kmers(infile, outfolder, threads=1, k=7, overwrite=False, verbose=False):
    """Count k-mers in a FASTQ file (drop-in for commands/image.py:727-806).

    Writes forward-strand counts to `<name minus suffixes>+k<k>.fq.h5` in outfolder
    (same file name rule, :752-759; the content is this package's own container, it
    is only ever read back by make_image).  Returns OrderedDict
    {"<k>mer_counting_time": seconds}, or an empty one when the file exists and
    overwrite is False (:761-763).  Raises on failure like the reference's
    check=True subprocess does; `threads` is accepted for signature parity.
    """
    import pandas as pd
    start_time = pd.Timestamp.now()

    Path(outfolder).mkdir(exist_ok=True)
    outfile = (str(Path(infile).name.removesuffix("".join(Path(infile).suffixes)))
               + BP_KMER_SEP + "k" + str(k) + ".fq.h5")
    outpath = Path(outfolder) / outfile

    if not overwrite and outpath.is_file():
        eprint("File exists. Skipping kmer counting for file:", str(infile))
        return OrderedDict()

    data = read_fastq_bytes(infile)
    eng = _engine(k, "count")
    hist, status = eng.count_host(data)
    if status:
        raise RuntimeError(f"k-mer counting failed for {infile}: inconsistent FASTQ framing "
                           f"(status bits {status})")
    if verbose:
        eprint(f"vk_count_host k={k} bytes={len(data)} windows={int(hist.sum(dtype=np.uint64))}")
    write_counts(outpath, k, hist)

    done_time = pd.Timestamp.now()
    stats = OrderedDict()
    stats[str(k) + "mer_counting_time"] = (done_time - start_time).total_seconds()
    return stats


def image_array(hist, kmer_mapping):
    """uint8 [side, side] image of one forward-strand histogram for a reference-style
    mapping DataFrame: the arithmetic of commands/image.py:897-919 on the GPU."""
    k, lut, npix = lut_from_dataframe(kmer_mapping)
    if hist.size != 4 ** k:
        raise IndexError("k-mer counts do not match the k-mer mapping size")
    key = hashlib.sha1(np.ascontiguousarray(lut).tobytes()).hexdigest()
    eng = _engine(k, key, lut=lut, npix=npix)
    return eng.image_host(hist)


def make_image(infile, outfolder, kmer_mapping, threads=1, overwrite=False, verbose=False, labels=[],
               base_sd=0, base_sd_thresh=QUAL_THRESH, subfolder_levels=0, mapping_code="varKode"):
    """Create an image from k-mer counts (drop-in for commands/image.py:808-936).

    Output name `<sample>@<bp>K+<mapping>+k<k>.png` (:840-849), optional md5-hex
    sharding folders (:850-854), skip-if-exists returning an empty OrderedDict
    (:857-859), 8-bit "L" PNG with the four varkoder* text chunks in the reference's
    order (:923-927), and {"k<k>_img_time": seconds} (:932-936).  A counts file with
    no k-mers raises pandas' EmptyDataError exactly as the reference's read_csv of an
    empty dsk2ascii dump does (:897-899).
    """
    import pandas as pd
    from PIL import Image
    from PIL.PngImagePlugin import PngInfo

    in_basename = str(Path(infile).name.removesuffix("".join(Path(infile).suffixes)))
    in_base1, in_k = in_basename.split(BP_KMER_SEP)

    outfile = in_base1 + BP_KMER_SEP + mapping_code + BP_KMER_SEP + in_k + ".png"
    outfolder = Path(outfolder)
    if subfolder_levels:
        hsh = list(hashlib.md5(outfile.encode("UTF-8")).hexdigest())
        for i in range(subfolder_levels):
            outfolder = outfolder / hsh.pop()
    Path(outfolder).mkdir(exist_ok=True, parents=True)

    if not overwrite and (outfolder / outfile).is_file():
        eprint("File exists. Skipping image for file:", str(infile))
        return OrderedDict()

    start_time = pd.Timestamp.now()
    kmer_size = len(kmer_mapping.index[0])

    k_file, hist = read_counts(infile)
    if k_file != kmer_size:
        raise IndexError(f"{infile}: counted with k={k_file} but the mapping has k={kmer_size}")
    if not hist.any():
        raise pd.errors.EmptyDataError("No columns to parse from file")

    kmer_array = image_array(hist, kmer_mapping)
    img = Image.fromarray(kmer_array)  # uint8 2-D -> mode "L"

    metadata = PngInfo()
    metadata.add_text("varkoderKeywords", LABELS_SEP.join(labels))
    metadata.add_text("varkoderBaseFreqSd", str(base_sd))
    metadata.add_text("varkoderLowQualityFlag", str(base_sd > base_sd_thresh))
    metadata.add_text("varkoderMapping", mapping_code)

    img.save(Path(outfolder) / outfile, optimize=True, pnginfo=metadata)

    done_time = pd.Timestamp.now()
    stats = OrderedDict()
    stats["k" + str(kmer_size) + "_img_time"] = (done_time - start_time).total_seconds()
    return stats
