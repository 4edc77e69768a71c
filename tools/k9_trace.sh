#!/bin/bash
# kernel-trace averages of the k=9 count for a list of builds: bash tools/k9_trace.sh default ab/x.so ...   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  OUT=gpurun_out/r05/trace_$tag
  mkdir -p $OUT
  if [ "$lib" = "default" ]; then unset VKIMG_LIB; else export VKIMG_LIB=$lib; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/k9_once.py > $OUT/out.txt 2>&1
  python3 - "$OUT" "$tag" <<'PY'
import csv, sys
print(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1] + "/t_kernel_stats.csv")):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith("vk_") and "synth" not in n and "lut" not in n:
        print("   %-40s %3s x %8.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
