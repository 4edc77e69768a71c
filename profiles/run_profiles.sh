#!/bin/bash
# Collect the rocprofv3 evidence for one round.  Usage (on the GPU box, from the repo root):
#   bash profiles/run_profiles.sh r01
# Kernel trace/stats and each PMC group run as separate passes (gpurun refuses mixes).
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --no-config4 --no-realistic --no-ladder --no-query --no-live-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
BENCHS="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-config4 --no-realistic --no-ladder --no-query --no-live-traffic"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCHS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $BENCHS > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq1 -o pmc -- $BENCHS > $OUT/bench_sq1.json 2> $OUT/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o pmc -- $BENCHS > $OUT/bench_sq2.json 2> $OUT/sq2.err
# fastp-shaped reads (dist 2): the count kernel and vk_aside_kernel over 6 launches
BENCH2="python3 bench.py --dist 2 --pool 256 --steps 6 --warmup 1 --no-cpu-baseline --no-e2e --no-config4 --no-realistic --no-ladder --no-query --no-live-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_d2 -o trace -- $BENCH2 > $OUT/bench_trace_d2.json 2> $OUT/trace_d2.err
# the 1-2-5 ladder of 32 samples (index + walker kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ladder -o trace -- python3 tools/ladder_time.py 32 1000000 7 > $OUT/ladder.txt 2> $OUT/trace_ladder.err
find $OUT -name "*.csv" | head -40
