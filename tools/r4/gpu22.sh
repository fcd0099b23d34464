#!/bin/bash
# round 4, GPU call 22: full GPU suite + the default bench line of the final code (with workloads and the CPU baseline)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
echo "== full GPU suite"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tail -12
echo "== default bench (shipped lib) with workloads"
timeout 1500 python bench.py 2>gpurun_out/r4/bench22.err | tail -1 > gpurun_out/r4/bench22.json; python - <<'P'
import json
d=json.load(open('gpurun_out/r4/bench22.json'))
print({k:d.get(k) for k in ('value','ms_per_step','eager_value','eager_ms_per_step')}); print(d['config'].get('pairs_per_gpu_per_step'), d['config'].get('sub_batch_streams'), d['config'].get('latency_ms_1_pair'))
print({k:d['roofline'].get(k) for k in ('kernel','achieved','frac','avg_launch_ms','traffic')})
for r in d['rooflines']: print({k:r.get(k) for k in ('kernel','achieved','frac','avg_launch_ms','traffic')})
print(d['config'].get('stage_ms_per_step'))
print(d['cpu_baseline'].get('value'), d['cpu_baseline'].get('pytorch_rocm_eager_same_gpu',{}).get('value'), d.get('other_precision',{}).get('value'))
for k,v in d.get('workloads',{}).items(): print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','eager_value','error','skipped')}, (v.get('pytorch_rocm_eager_same_gpu') or {}).get('value'))
P
tail -3 gpurun_out/r4/bench22.err
