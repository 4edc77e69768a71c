"""Constants of the drop-in surface (reference: varKoder/core/config.py:18-24, 33-34)."""

# file naming (core/config.py:18-21)
LABEL_SAMPLE_SEP = "+"
LABELS_SEP = ";"
BP_KMER_SEP = "+"
SAMPLE_BP_SEP = "@"

# quality flag threshold (core/config.py:24)
QUAL_THRESH = 0.01

# k-mer mapping (core/config.py:27, 33-34)
MAPPING_CHOICES = ["varKode", "cgr"]
DEFAULT_KMER_SIZE = 7
DEFAULT_KMER_MAPPING = "cgr"

KMER_MIN, KMER_MAX = 5, 9  # commands/image.py:1209-1210
