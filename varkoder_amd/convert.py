"""Mirror of `varKoder convert`'s remap() (reference: varKoder/commands/convert.py:34-77):
re-lay an image from one k-mer mapping to the other.  The reference joins the two mapping
tables on the k-mer string and scatters pixel by pixel; here the join is folded once into a
source-pixel map per (k, direction) and the scatter is a gather kernel behind vk_remap_host.

What the reference's join + NumPy assignment amounts to (pinned by tests/golden, which hold
outputs of the reference's own remap() on asymmetric images):
  * varKode -> cgr: cgr pixels of s and of rc(s) both read the varKode pixel of the class.
  * cgr -> varKode: the four candidate writes into a class's pixel come in the row order of
    get_cgr's table; the last one wins, i.e. the pixel reads the cgr pixel of whichever of
    {s, rc(s)} is later in get_cgr's enumeration (core/utils.py:188: meshgrid order, base 1
    fastest, then base 0, then bases 2..k-1).  Cells without a k-mer stay 0.
  * cgr -> varKode with sum_rc: np.add.at on a uint8 array: 2*cgr(s) + 2*cgr(rc s) mod 256
    (4*cgr(s) for a palindrome, whose row is duplicated in the shipped table), then
    uint8((v - min) / max * 255) in float64.
"""
import ctypes as C
import functools

import numpy as np

from . import _capi
from .config import MAPPING_CHOICES
from .mapping import pixel_lut, revcomp_codes, side

UNMAPPED = np.uint32(0xFFFFFFFF)


def cgr_row_order(k):
    """Position of every code in get_cgr's all_sequences enumeration."""
    codes = np.arange(4 ** k, dtype=np.int64)
    b = [(codes >> (2 * (k - 1 - i))) & 3 for i in range(k)]
    digits = [b[1], b[0]] + b[2:] if k >= 2 else b
    order = np.zeros_like(codes)
    for w, dgt in enumerate(digits):
        order += dgt * (4 ** w)
    return order


@functools.lru_cache(maxsize=None)
def source_maps(k, in_mapping, out_mapping):
    """(src0, src1, w0, w1, npix_in, npix_out) for vk_remap_host."""
    lin, lout = pixel_lut(k, in_mapping).astype(np.int64), pixel_lut(k, out_mapping).astype(np.int64)
    nin, nout = side(k, in_mapping) ** 2, side(k, out_mapping) ** 2
    codes = np.arange(4 ** k)
    rc = revcomp_codes(k).astype(np.int64)
    src0 = np.full(nout, UNMAPPED, dtype=np.uint32)
    src1 = np.full(nout, UNMAPPED, dtype=np.uint32)
    w0 = np.zeros(nout, dtype=np.uint8)
    w1 = np.zeros(nout, dtype=np.uint8)
    if in_mapping == "cgr" and out_mapping == "varKode":
        order = cgr_row_order(k)
        win = np.where(order >= order[rc], codes, rc)
        src0[lout[codes]] = lin[win]
        src1[lout[codes]] = lin[rc[win]]
        pal = rc == codes
        w0[lout[codes]] = np.where(pal, 4, 2)
        w1[lout[codes]] = np.where(pal, 0, 2)
    else:
        src0[lout[codes]] = lin[codes]
        # summing is only meaningful cgr -> varKode; other directions add each source once per
        # joined row: varKode -> cgr joins one varKode row with two cgr rows per k-mer
        src1[lout[codes]] = lin[codes]
        w0[lout[codes]] = 1
        w1[lout[codes]] = 0
    for a in (src0, src1, w0, w1):
        a.setflags(write=False)
    return src0, src1, w0, w1, nin, nout


def remap_array(arr, k, in_mapping, out_mapping, sum_rc=False, engine=None):
    """uint8 [side_in, side_in] (or a batch [n, side_in, side_in]) -> remapped array(s)."""
    if (in_mapping not in MAPPING_CHOICES) or (out_mapping not in MAPPING_CHOICES):
        raise Exception("Input and output mapping must be one of: " + str(MAPPING_CHOICES))
    if sum_rc and not (in_mapping == "cgr" and out_mapping == "varKode"):
        raise NotImplementedError("sum_rc is supported for cgr -> varKode only")
    src0, src1, w0, w1, nin, nout = source_maps(k, in_mapping, out_mapping)
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    batch = a.reshape(-1, nin)
    out = np.empty((batch.shape[0], nout), dtype=np.uint8)
    if engine is None:
        from .image import _engine
        engine = _engine(k, "count")
    vp = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731
    st = engine.L.vk_remap_host(engine.ctx, vp(batch), batch.shape[0], nin, nout, vp(src0), vp(src1), vp(w0),
                                vp(w1), 1 if sum_rc else 0, vp(out))
    _capi.check(engine.ctx, st, "vk_remap_host")
    n = side(k, out_mapping)
    return out.reshape((n, n) if a.ndim == 2 else (-1, n, n))


def remap(img, k, in_mapping, out_mapping, sum_rc=False):
    """Drop-in for convert.py:34-77: PIL image in, PIL image out."""
    from PIL import Image
    return Image.fromarray(remap_array(np.array(img), k, in_mapping, out_mapping, sum_rc))


# ---- the `convert` command at folder level (commands/convert.py:80-202) -----------------------

def get_metadata_from_img_filename(img_path):
    """`<sample>@<bp>K+<mapping>+k<k>.png` -> dict (core/utils.py:123-147); a legacy two-field
    name `<sample>@<bp>K+k<k>.png` means mapping 'varKode'.  Raises ValueError on anything else."""
    from pathlib import Path

    from .config import BP_KMER_SEP, SAMPLE_BP_SEP
    sample_name, rest = Path(img_path).name.removesuffix(".png").split(SAMPLE_BP_SEP)
    fields = rest.split(BP_KMER_SEP)
    if len(fields) == 3:
        n_bp, mapping, ksize = fields
    else:
        n_bp, ksize = fields          # ValueError unless exactly two
        mapping = "varKode"
    return {"sample": sample_name, "bp": int(n_bp[:-1]) * 1000, "img_kmer_mapping": mapping,
            "img_kmer_size": int(ksize[1:]), "path": Path(img_path)}


def plan_conversion(input_dir, outdir, output_mapping, input_mapping=None, kmer_size=None):
    """One record per *.png under input_dir with the reference's naming rules
    (ConvertCommand._collect_image_files, convert.py:135-176), kept as they are:
    -p / -k given on the command line override what the file name says; the output keeps the
    sub-folders of the input minus the first level; a file whose name does not parse keeps its
    path minus its first component."""
    from pathlib import Path

    from .config import BP_KMER_SEP, SAMPLE_BP_SEP
    records = []
    for f in Path(input_dir).rglob("*.png"):
        try:
            md = get_metadata_from_img_filename(f)
            if input_mapping:
                md["img_kmer_mapping"] = input_mapping
            if kmer_size:
                md["img_kmer_size"] = kmer_size
        except Exception:  # noqa: BLE001 - as the reference: any parse failure
            md = {"sample": None, "bp": None, "img_kmer_mapping": input_mapping, "img_kmer_size": kmer_size, "path": f}
        if md["sample"] and md["bp"]:
            fname = (f"{md['sample']}{SAMPLE_BP_SEP}{int(md['bp'] / 1000):08d}K{BP_KMER_SEP}"
                     f"{output_mapping}{BP_KMER_SEP}k{md['img_kmer_size']}.png")
            md["outfile_path"] = Path(outdir) / Path(*f.relative_to(Path(input_dir)).parent.parts[1:]) / fname
        else:
            md["outfile_path"] = Path(outdir) / Path(*f.parts[1:])
        records.append(md)
    return records


def convert_folder(input_dir, outdir, output_mapping, input_mapping=None, kmer_size=None, sum_rc=False,
                   overwrite=False, io_threads=8):
    """`varKoder convert` for a folder of images: all images of one (k, input mapping) go through
    ONE vk_remap_host call; PNG text chunks are carried over with `varkoderMapping` updated when
    present (convert.py:108-121).  Returns the number of images written."""
    import os
    from collections import defaultdict
    from concurrent.futures import ThreadPoolExecutor
    from pathlib import Path

    from PIL import Image
    from PIL.PngImagePlugin import PngInfo
    if not overwrite and Path(outdir).exists():
        raise Exception("Output directory exists, use --overwrite if you want to overwrite it.")
    groups = defaultdict(list)
    for md in plan_conversion(input_dir, outdir, output_mapping, input_mapping, kmer_size):
        if md["img_kmer_mapping"] == output_mapping:
            continue                                                     # convert.py:88-89
        if os.path.exists(md["outfile_path"]) and not os.access(md["outfile_path"], os.W_OK):
            continue                                                     # convert.py:92-93
        groups[(md["img_kmer_size"], md["img_kmer_mapping"])].append(md)

    def save(md, arr, info):
        chunks = PngInfo()
        for key, value in info.items():
            chunks.add_text(key, output_mapping if key == "varkoderMapping" else str(value))
        md["outfile_path"].parent.mkdir(parents=True, exist_ok=True)
        Image.fromarray(arr).save(md["outfile_path"], optimize=True, pnginfo=chunks)

    written = 0
    with ThreadPoolExecutor(io_threads) as pool:
        for (k, in_mapping), mds in groups.items():
            if (in_mapping not in MAPPING_CHOICES) or (output_mapping not in MAPPING_CHOICES):
                raise Exception("Input and output mapping must be one of: " + str(MAPPING_CHOICES))
            opened = list(pool.map(lambda md: Image.open(md["path"]), mds))
            batch = np.stack([np.array(im) for im in opened])
            out = remap_array(batch, k, in_mapping, output_mapping, sum_rc)
            list(pool.map(lambda t: save(t[0], out[t[1]], opened[t[1]].info), [(md, i) for i, md in enumerate(mds)]))
            written += len(mds)
    return written
