#!/bin/bash
# Instructions per 4 KiB piece of pass A (vk_bucket_kernel<9, 0>) for the shipped build and for diagnostic builds that
# leave a stage out (tools/build_rev.sh WORK k9_nodrain -DVK_DIAG_K9_NO_DRAIN, ..._NO_APPEND, ..._NO_SINGLES):
#   bash tools/k9_stage_pmc.sh default ab/k9_nodrain.so ab/k9_noappend.so ab/k9_nosingles.so      (GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  OUT=gpurun_out/r05/k9stage_$tag
  mkdir -p $OUT
  if [ "$lib" = "default" ]; then unset VKIMG_LIB; else export VKIMG_LIB=$lib; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT -o pmc -- python3 bench.py --k 9 --mapping cgr --samples 100 --pool 100 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-live-traffic > $OUT/bench.json 2> $OUT/err.txt
  python3 - "$OUT" "$tag" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1] + "/pmc_counter_collection.csv")):
    if "vk_bucket_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
pieces = 100 * 320e6 / 4096
print("%-14s" % sys.argv[2], {k: round(sum(v) / len(v) / pieces, 1) for k, v in sorted(agg.items())})
PY
done
