#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench workload (timed region only, no graph): bash tools/prof_workload.sh <workload> <tag> [extra bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$1; TAG=$2; shift 2
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --workload $W --timed-only --no-graph --steps 5 --warmup 2 "$@" > $OUT/stdout.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
ls $OUT
