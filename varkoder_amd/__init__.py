"""varkoder_amd -- MI355X-native implementation of varKoder's `image` hot path.

FASTQ reads -> canonical k-mer counts -> varKode / rfCGR image, as hand-written
HIP kernels for gfx950 behind a C ABI (include/vkimg.h), with a host-side mirror
of the reference's Python interface (varKoder/commands/image.py:727-936).
"""
from .config import (BP_KMER_SEP, DEFAULT_KMER_MAPPING, DEFAULT_KMER_SIZE, LABELS_SEP, MAPPING_CHOICES,
                     QUAL_THRESH, SAMPLE_BP_SEP)
from .mapping import get_cgr, get_kmer_mapping, pixel_lut

__all__ = ["get_kmer_mapping", "get_cgr", "pixel_lut", "count_kmers", "make_image", "ImageEngine",
           "BP_KMER_SEP", "SAMPLE_BP_SEP", "LABELS_SEP", "QUAL_THRESH", "DEFAULT_KMER_SIZE",
           "DEFAULT_KMER_MAPPING", "MAPPING_CHOICES"]

__version__ = "0.1.0"


def __getattr__(name):
    # image/engine pull in PIL/torch; keep `import varkoder_amd` light
    if name in ("count_kmers", "make_image"):
        from . import image
        return getattr(image, name)
    if name == "ImageEngine":
        from .engine import ImageEngine
        return ImageEngine
    raise AttributeError(name)
