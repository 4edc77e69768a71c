#!/bin/bash
# PMC passes for the k=9 spill path (BASELINE config 4): bash profiles/run_k9_pmc.sh <tag>
set -e
TAG=${1:-k9}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --k 9 --mapping cgr --samples 100 --pool 100 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-live-traffic $K9_EXTRA"
# (the kernel-trace pass over 12 launches of every kernel: an average worth quoting; the counter passes over one)
BT="python3 bench.py --k 9 --mapping cgr --samples 100 --pool 100 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-live-traffic $K9_EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BT > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq1 -o pmc -- $B > $OUT/bench_sq1.json 2> $OUT/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o pmc -- $B > $OUT/bench_sq2.json 2> $OUT/sq2.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $B > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $B > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head
