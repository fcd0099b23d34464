#!/bin/bash
# round 4, GPU call 17: sub-batch streams x batch with the final kernels (timed only, interleaved, 2 repetitions)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for cfg in "--streams 1 --batch 8" "--streams 2 --batch 8" "--streams 3 --batch 9" "--streams 2 --batch 12" "--streams 4 --batch 8"; do
  v=$(python bench.py --timed-only --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$cfg => $v"
done
done
