# whole-model throughput by (pairs per step, sub-batch streams): bash tools/bench_batch_streams.sh
for cfg in "8 2" "16 2" "12 3" "16 4" "12 2" "8 2"; do
  set -- $cfg
  v=$(python bench.py --timed-only --steps 12 --warmup 4 --batch $1 --streams $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "batch $1 streams $2 => $v"
done
