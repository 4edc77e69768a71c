#!/bin/bash
# Vector / LDS instruction counts and LDS conflict cycles of the count kernels on low-complexity reads:
#   bash tools/repeat_pmc.sh <tag> <k> <kind>     (GPU box; two rocprofv3 --pmc passes of tools/repeat_once.py)
set -e
TAG=${1:-x}; K=${2:-7}; KIND=${3:-acgt}
OUT=gpurun_out/rep_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o pmc -- python3 tools/repeat_once.py $K $KIND > $OUT/a.txt 2> $OUT/a.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -o pmc -- python3 tools/repeat_once.py $K $KIND > $OUT/b.txt 2> $OUT/b.err
python3 - "$OUT" <<'PY'
import csv, sys, collections
pieces = 512 * 64e6 / 4096
for sub in ("a", "b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"{sys.argv[1]}/{sub}/pmc_counter_collection.csv")):
        kn = r["Kernel_Name"]; name = kn[kn.find("vk_"):].split("(")[0].split("<")[0] if "vk_" in kn else kn
        if "count" in name or "bucket" in name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in agg.items():
        print(name, {c: round(sum(v) / len(v) / pieces, 1) for c, v in cs.items()}, "(per 4 KiB piece)")
PY
cat $OUT/a.txt | grep -v amdgpu
