#!/bin/bash
# kernel-trace averages of the count kernels on low-complexity reads: bash tools/repeat_trace.sh <k> <kind> [<k> <kind> ...]   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
while [ $# -ge 2 ]; do
  K=$1; KIND=$2; shift 2
  OUT=gpurun_out/r05/rtrace_${K}_$KIND
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/repeat_once.py $K $KIND > $OUT/out.txt 2>&1
  grep "^k=" $OUT/out.txt
  python3 - "$OUT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + "/t_kernel_stats.csv")):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith("vk_") and "synth" not in n and "lut" not in n:
        print("   %-40s %3s x %8.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
