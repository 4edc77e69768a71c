#!/bin/bash
# Kernel trace + SQ counters of vk_inflate_device on 64 gzip files of 128 MB of FASTQ text each:
#   bash profiles/run_inflate.sh <tag> [level]
set -e
TAG=${1:-inflate}
LEVEL=${2:-6}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 tools/inflate_time.py 64 400000 $LEVEL"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B > $OUT/run_trace.log 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_sq1 -o pmc -- $B > $OUT/run_sq1.log 2> $OUT/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o pmc -- $B > $OUT/run_sq2.log 2> $OUT/sq2.err
grep -h "GPU inflate" $OUT/run_*.log
