#!/bin/bash
# Round-5 final pass on the GPU box (through gpurun), everything at ONE code state:
#   1. the whole GPU suite                                   -> gpurun_out/r5_final/suite.log
#   2. tools/profile_round4.sh r5 (kernel stats + FETCH_SIZE / WRITE_SIZE / SQ passes of `bench.py --timed-only --no-graph --streams 1`)
#   3. the default bench line as the driver runs it          -> gpurun_out/r5_final/bench_default.json
#   4. replay-to-replay determinism of the timed configuration and of the head next to the marching kernel (tools/diag_*.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_final; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -v GridwiseOp > $O/suite.log; tail -4 $O/suite.log
bash tools/profile_round4.sh r5 > $O/profile.log 2>&1; tail -3 $O/profile.log
cd $GRAFT_REPO_ROOT
T0=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; T1=$(date +%s)
echo "python bench.py --gpus 1 --steps 20 --warmup 5: $((T1 - T0)) s wall, $(grep -c . $O/bench_default.json) line(s) on stdout" > $O/bench_time.txt; head -c 700 $O/bench_default.json; echo; cat $O/bench_time.txt
python tools/ctypes_census.py 2>&1 | grep "^\[\|extension" > $O/census.txt
for i in 1 2 3; do python tools/diag_timed_config.py --tag final_$i 2>&1 | grep "^\[" ; done > $O/determinism.txt
python tools/diag_head_under_load.py --load f16x3 --iters 60 --tag "shipped head next to the marching kernel" 2>&1 | grep "^\[" >> $O/determinism.txt
cat $O/determinism.txt
# 5. socket power / shader clock while the timed configuration runs (what power the headline assumes); parse_smi.py reads
#    smi_f16x3.jsonl + bench_f16x3.json of a directory
P=$O/power; mkdir -p $P
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.2; done ) > $P/smi_f16x3.jsonl & SMI=$!
timeout 200 python bench.py --timed-only --steps 400 --warmup 5 > $P/bench_f16x3.json 2>/dev/null
kill $SMI; wait $SMI 2>/dev/null
python tools/parse_smi.py $P > $O/power_summary.txt 2>&1; cat $O/power_summary.txt; rm -f $P/smi_f16x3.jsonl

