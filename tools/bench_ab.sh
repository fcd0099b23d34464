# interleaved whole-model A/B on one box: bash tools/bench_ab.sh "<env A>" "<env B>" ...   (each arg: space-separated VAR=VALUE list)
for rep in 1 2; do
  for cfg in "$@"; do
    v=$(env $cfg python bench.py --timed-only --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "$cfg => $v"
  done
done
