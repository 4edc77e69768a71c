#!/bin/bash
# kernel-trace averages of vk_inflate_device for a list of builds: bash tools/gz_trace.sh <level> default ab/x.so ...   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
LEVEL=$1; shift
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  OUT=gpurun_out/gztrace/$tag
  mkdir -p $OUT
  if [ "$lib" = "default" ]; then unset VKIMG_LIB; else export VKIMG_LIB=$lib; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/inflate_time.py 64 400000 $LEVEL > $OUT/out.txt 2>&1
  echo "$tag: $(grep 'GPU inflate' $OUT/out.txt | sed 's/.*GPU inflate/GPU inflate/' | cut -c1-60)"
  python3 - "$OUT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + "/t_kernel_stats.csv")):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith("vk_") and ("gz" in n or "crc" in n or "inflate" in n):
        print("   %-28s %3s x %8.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
