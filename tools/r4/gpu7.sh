#!/bin/bash
# round 4, GPU call 7: the whole GPU suite + the default bench line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
echo "== full GPU suite"; timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tail -30
echo "== default bench (shipped lib) with workloads"
timeout 1200 python bench.py 2>gpurun_out/r4/bench7.err | tail -1 > gpurun_out/r4/bench7.json; python - <<'P'
import json
d=json.load(open('gpurun_out/r4/bench7.json'))
print({k:d.get(k) for k in ('value','ms_per_step','eager_value','eager_ms_per_step')}); print(d['roofline']); print(d.get('cpu_baseline',{}).get('value'))
for k,v in d.get('workloads',{}).items(): print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','eager_value','error','skipped')})
P
tail -5 gpurun_out/r4/bench7.err
