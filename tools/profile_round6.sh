#!/bin/bash
# Round-6 final pass on the GPU box (through gpurun), everything at ONE code state:
#   1. tools/profile_round4.sh r6: rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE / SQ passes of `bench.py --timed-only --no-graph --streams 1`
#   2. the default bench line as the driver runs it          -> gpurun_out/r6_final/bench_default.json
#   3. replay-to-replay determinism of the timed configuration (tools/diag_timed_config.py x 3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_final; mkdir -p $O
git rev-parse HEAD > $O/commit.txt 2>/dev/null || true
SKIP_CAL=1 bash tools/profile_round4.sh r6 > $O/profile.log 2>&1; tail -3 $O/profile.log
cd $GRAFT_REPO_ROOT
T0=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; T1=$(date +%s)
echo "python bench.py --gpus 1 --steps 20 --warmup 5: $((T1 - T0)) s wall, $(grep -c . $O/bench_default.json) line(s) on stdout" > $O/bench_time.txt; head -c 900 $O/bench_default.json; echo; cat $O/bench_time.txt
for i in 1 2 3; do python tools/diag_timed_config.py --tag final_$i 2>&1 | grep "^\[" ; done > $O/determinism.txt; cat $O/determinism.txt
