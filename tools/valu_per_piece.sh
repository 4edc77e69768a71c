#!/bin/bash
# VALU / SALU / LDS instructions per 4 KiB piece of the k=7 count kernel of the library in the tree (or $VKIMG_LIB):
#   bash tools/valu_per_piece.sh <tag>      (GPU box; one rocprofv3 --pmc pass, 1000-sample launch)
#   VALU_EXTRA="--dist 2 --pool 256" adds bench.py arguments (the piece count follows the bench line's bytes per sample)
set -e
TAG=${1:-x}
OUT=gpurun_out/valu_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-config4 --no-realistic --no-ladder --no-query --no-live-traffic $VALU_EXTRA > $OUT/bench.json 2> $OUT/err.txt
python3 - "$OUT" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1] + "/pmc_counter_collection.csv")):
    if "vk_count" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
import json
line = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
pieces = line["roofline"]["algorithmic_bytes_per_launch"] / 4096
print("launch: %.2f ms, %.1f M pieces" % (line["roofline"]["avg_launch_ms"], pieces / 1e6))
print({k: round(sum(v) / len(v) / pieces, 1) for k, v in agg.items()})
PY
