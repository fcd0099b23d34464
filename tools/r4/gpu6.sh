#!/bin/bash
# round 4, GPU call 6: marching kernel v4 (async double-buffered LDS-DMA staging) + default bench with the amp workloads
cd "$(dirname "$0")/../.."
V=openstereo_amd/lib/variants
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "marching or split_activation or pipelined or gwcnet_small or gwc_disp_processor or psmnet_256" 2>&1 | tail -6
export OSA_PRECISION=f16x3
for B in 8 4 1; do
echo "== layers B=$B split chain"
OSA_LIB_PATH=$V/m4.so timeout 600 python tools/bench_layers.py --set 3d --batch $B --iters 10 --split --only "V0" \
   --envs "OSA_MARCH=0;OSA_MARCH_GEO=0;OSA_MARCH_GEO=1;OSA_MARCH_GEO=0,OSA_MARCH_NSEG=2" 2>&1 | grep -v "amdgpu.ids\|redir1\|classif"
done
echo "== layers B=8 fp32 in/out"
OSA_LIB_PATH=$V/m4.so timeout 600 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --only "V0" --envs "OSA_MARCH=0;OSA_MARCH_GEO=0;OSA_MARCH_GEO=1" 2>&1 | grep -v "amdgpu.ids\|redir1\|classif"
echo "== whole model A/B (timed only)"
OSA_LIB_PATH=$V/m4.so bash tools/bench_ab.sh "OSA_MARCH=0" "OSA_MARCH_GEO=0" "OSA_MARCH_GEO=1" 2>&1 | grep -v amdgpu.ids
unset OSA_PRECISION
echo "== default bench (shipped lib) with workloads"
timeout 1200 python bench.py --no-cpu-baseline 2>gpurun_out/r4/bench6.err | tail -1 > gpurun_out/r4/bench6.json; python - <<'P'
import json
d=json.load(open('gpurun_out/r4/bench6.json'))
print({k:d.get(k) for k in ('value','ms_per_step','eager_value','eager_ms_per_step')}); print(d['roofline'])
for k,v in d.get('workloads',{}).items(): print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','eager_value','error','skipped','dtype')})
P
tail -5 gpurun_out/r4/bench6.err
