#!/bin/bash
# Round-4 profile (GPU box, through gpurun; summaries are copied to profiles/round4/):
#   1. rocprofv3 --kernel-trace --stats of `bench.py --timed-only --no-graph` at the default batch (9 pairs per step since late r4; BATCH=8 reproduces the earlier rounds)
#   2. separate --pmc passes: FETCH_SIZE, WRITE_SIZE (HBM bytes per launch), and one SQ pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
#      SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY) + GRBM_GUI_ACTIVE -- MFMA utilisation of every conv instance
#   3. the same FETCH_SIZE / WRITE_SIZE passes over tools/calibrate_traffic.py (known byte counts in the engine's own access patterns;
#      SKIP_CAL=1 skips them -- the calibration does not depend on the kernels)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r4}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --timed-only --no-graph --streams 1 --steps 5 --warmup 2 --batch ${BATCH:-9}"   # one stream: per-kernel durations undisturbed (the default run issues every launch as 2 concurrent half-batches)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace_stdout.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- $CMD > $OUT/pmc_${C}_stdout.log 2>&1
  [ -z "$WITH_CAL" ] || timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/cal_$C -o p -- python $R/tools/calibrate_traffic.py > $OUT/cal_${C}_stdout.log 2>&1
done
[ -n "$SKIP_SQ" ] || timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_SQ -o p -- $CMD > $OUT/pmc_SQ_stdout.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python $R/tools/parse_pmc3.py $OUT ${BATCH:-9} > $OUT/traffic.json 2> $OUT/parse.log
ls $OUT
