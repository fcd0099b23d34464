#!/bin/bash
# round 4, GPU call 1: marching-form parity + per-layer A/B (experiments build) + whole-model A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
export OSA_PRECISION=f16x3
EXP=openstereo_amd/lib/variants/exp.so
echo "== parity" ; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "marching or conv3d_bn_act or split_activation or pipelined or gwcnet_small or gwc_disp_processor or gwc_hourglass" 2>&1 | tail -15
echo "== layers B=8 split chain (brick vs marching geometries)"
OSA_LIB_PATH=$EXP timeout 600 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --split --only "V0" \
   --envs "OSA_MARCH=0;OSA_MARCH_GEO=0;OSA_MARCH_GEO=1;OSA_MARCH_GEO=2;OSA_MARCH_GEO=0,OSA_MARCH_NSEG=2;OSA_MARCH_GEO=1,OSA_MARCH_NSEG=2" 2>&1 | grep -v amdgpu.ids
echo "== layers B=4 split chain"
OSA_LIB_PATH=$EXP timeout 600 python tools/bench_layers.py --set 3d --batch 4 --iters 10 --split --only "V0" \
   --envs "OSA_MARCH=0;OSA_MARCH_GEO=0;OSA_MARCH_GEO=1;OSA_MARCH_GEO=2;OSA_MARCH_GEO=0,OSA_MARCH_NSEG=2;OSA_MARCH_GEO=1,OSA_MARCH_NSEG=3" 2>&1 | grep -v amdgpu.ids
echo "== layers B=1 split chain"
OSA_LIB_PATH=$EXP timeout 600 python tools/bench_layers.py --set 3d --batch 1 --iters 10 --split --only "V0" \
   --envs "OSA_MARCH=0;OSA_MARCH_GEO=0;OSA_MARCH_GEO=1;OSA_MARCH_GEO=0,OSA_MARCH_NSEG=2;OSA_MARCH_GEO=0,OSA_MARCH_NSEG=8" 2>&1 | grep -v amdgpu.ids
echo "== layers B=8 fp32 in/out (dres0.0-like staging with the split at staging)"
OSA_LIB_PATH=$EXP timeout 600 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --only "V0" \
   --envs "OSA_MARCH=0;OSA_MARCH_GEO=0;OSA_MARCH_GEO=1" 2>&1 | grep -v amdgpu.ids
echo "== whole model A/B (timed only)"
OSA_LIB_PATH=$EXP bash tools/bench_ab.sh "OSA_MARCH=0" "OSA_MARCH_GEO=0" "OSA_MARCH_GEO=1" 2>&1 | grep -v amdgpu.ids
echo "== default bench (shipped lib), eager figure included"
timeout 900 python bench.py --no-cpu-baseline --no-workloads 2>gpurun_out/r4/bench1.err | tail -1 > gpurun_out/r4/bench1.json; python - <<'P'
import json
d=json.load(open('gpurun_out/r4/bench1.json'))
print({k:d.get(k) for k in ('value','ms_per_step','eager_value','eager_ms_per_step')}); print(d['roofline']); print(d['config'].get('stage_ms_per_step'))
P
