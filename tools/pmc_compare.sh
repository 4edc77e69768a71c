#!/bin/bash
# Wave-time counters of the k<=7 count kernel for one bench configuration, per 4 KiB piece:
#   PMC_EXTRA="--dist 2 --pool 256" bash tools/pmc_compare.sh <tag>     (GPU box; separate rocprofv3 --pmc passes)
set -e
TAG=${1:-x}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-config4 --no-realistic --no-ladder --no-query --no-live-traffic $PMC_EXTRA"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT -o p1 -- python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/err1.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o p2 -- python3 bench.py $ARGS > /dev/null 2> $OUT/err2.txt
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH --output-format csv -d $OUT -o p3 -- python3 bench.py $ARGS > /dev/null 2> $OUT/err3.txt
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT --output-format csv -d $OUT -o p4 -- python3 bench.py $ARGS > /dev/null 2> $OUT/err4.txt
python3 - "$OUT" <<'PY'
import csv, sys, collections, json, glob
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "vk_count" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
line = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
pieces = line["roofline"]["algorithmic_bytes_per_launch"] / 4096
print("launch: %.2f ms, %.1f M pieces" % (line["roofline"]["avg_launch_ms"], pieces / 1e6))
for k in sorted(agg):
    print("  %-24s %10.1f per piece" % (k, sum(agg[k]) / len(agg[k]) / pieces))
PY
