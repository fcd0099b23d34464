#!/bin/bash
# steady-state kernel table of a training workload: bash tools/prof_train.sh <workload> <tag> <window_ms> <steps_in_window> [extra bench.py flags, e.g. --amp]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$1; TAG=$2; WIN=$3; NS=$4; EXTRA="${@:5}"
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $R/bench.py --workload $W --timed-only --no-graph --steps 12 --warmup 4 $EXTRA > $OUT/stdout.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/steady_state.py $F $WIN $NS 60 > $OUT/steady_state.txt 2>&1
rm -rf $OUT/trace
cat $OUT/steady_state.txt
