#!/bin/bash
# Start-up stagger sweep (experiments build): per-layer timings with the stagger off / on at several step scales.
# (the shipped library has the stagger compiled out of the launch path: OSA_STAG defaults to 0 and is only read in the experiments build)
export OSA_LIB_PATH=openstereo_amd/lib/variants/exp.so OSA_PRECISION=f16x3
B=${1:-8}
for setting in "OSA_STAG=0" "OSA_STAG=1,OSA_STAG_PCT=50" "OSA_STAG=1,OSA_STAG_PCT=100" "OSA_STAG=1,OSA_STAG_PCT=150" "OSA_STAG=1,OSA_STAG_PCT=250"; do
  echo "=== $setting (batch $B)"
  python tools/bench_layers.py --set 3d --batch $B --iters 10 --env "$setting" 2>&1 | grep -v amdgpu.ids
  python tools/bench_layers.py --set 2d --batch $B --iters 10 --env "$setting" 2>&1 | grep -v amdgpu.ids
done
OSA_STAG_PRINT=1 python tools/bench_layers.py --set 3d --batch $B --iters 1 2>&1 | grep stagger | sort | uniq -c
OSA_STAG_PRINT=1 python tools/bench_layers.py --set 2d --batch $B --iters 1 2>&1 | grep stagger | sort | uniq -c
