#!/bin/bash
# round 4, GPU call 8: what do the per-wave B (weight) loads cost in the BRICK kernel?  exp.so vs brick_nob.so (no per-tap B loads; timing only)
cd "$(dirname "$0")/../.."
V=openstereo_amd/lib/variants
export OSA_PRECISION=f16x3 OSA_MARCH=0
for lib in exp brick_nob; do
echo "== $lib: 3-D layers B=8"
OSA_LIB_PATH=$V/$lib.so timeout 600 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --dbgs 9 2>&1 | grep -v "amdgpu.ids"
echo "== $lib: 2-D layers B=8"
OSA_LIB_PATH=$V/$lib.so timeout 600 python tools/bench_layers.py --set 2d --batch 8 --iters 10 --dbgs 9 2>&1 | grep -v "amdgpu.ids"
done
echo "== amax kernel in the captured GwcNet training step (r3: NaN from the 2nd replay on)"
unset OSA_MARCH
OSA_ENGINE_AMAX=1 timeout 600 python bench.py --workload gwcnet_train --steps 10 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-600
