#!/bin/bash
# Counters of every kernel of one k=9 count launch (100 samples x 1M x 150 bp), per kernel: bash tools/k9q_pmc.sh <tag> [dist]   (GPU box)
set -e
TAG=${1:-k9q}; DIST=${2:-0}
OUT=gpurun_out/r05/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --k 9 --mapping cgr --samples 100 --pool 100 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-live-traffic --dist $DIST"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/a -o pmc -- $B > $OUT/bench_a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/b -o pmc -- $B > $OUT/bench_b.json 2> $OUT/b.err
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
        if k.startswith("vk_") and "synth" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
pieces = 100 * 320e6 / 4096
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print("   %-22s %14.0f   per piece %9.1f" % (c, m, m / pieces))
PY
