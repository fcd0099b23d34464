#!/bin/bash
# round 4, GPU call 18: sub-batch streams x batch, second sweep
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for cfg in "--streams 2 --batch 8" "--streams 3 --batch 9" "--streams 3 --batch 12" "--streams 4 --batch 12" "--streams 4 --batch 16" "--streams 3 --batch 6" "--streams 6 --batch 12"; do
  v=$(python bench.py --timed-only --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$cfg => $v"
done
done
