#!/bin/bash
# round 4, GPU call 27: tile 9 takes the weight ring for long K loops (>= 16 chunks) under the default mask: GRU layers, backbone check, IGEV / StereoBase workloads
# RECORD ONLY: the long-K rule lost 1.3 % on the IGEV loop and was reverted -- profiles/round4/b_ring_tile9_long_k.txt.
cd "$(dirname "$0")/../.."
export OSA_PRECISION=f16x3
echo "== gru / backbone layers: explicit default mask (no tile 9) vs built-in default (tile 9 for long K)"
for M in 8222 ""; do
  echo "-- OSA_B_RING_MASK=$M"
  OSA_B_RING_MASK=$M timeout 600 python tools/bench_layers.py --set gru --batch 4 --iters 10 --only "@136x240" 2>&1 | grep -v "amdgpu.ids\|^sum"
  OSA_B_RING_MASK=$M timeout 600 python tools/bench_layers.py --set 2d --batch 3 --iters 10 --only "128" 2>&1 | grep -v "amdgpu.ids\|^sum"
done
unset OSA_PRECISION
echo "== workloads"
for rep in 1 2; do
for M in 8222 ""; do
  for wl in igev_refine32 stereobase_e2e; do
    v=$(OSA_B_RING_MASK=$M python bench.py --workload $wl --timed-only --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "mask=$M $wl => $v"
  done
done
done
