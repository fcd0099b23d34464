#!/bin/bash
# round 4, GPU call 26: after the split builder's step change: parity subset + the default bench line of the final code
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py tests/test_torch_ext.py -q -m gpu -k "volume or gwcnet or at_size or torch_ext or gwc_" 2>&1 | tail -4
echo "== default bench (shipped lib) with workloads"
timeout 1500 python bench.py 2>gpurun_out/r4/bench26.err | tail -1 > gpurun_out/r4/bench26.json; python - <<'P'
import json
d=json.load(open('gpurun_out/r4/bench26.json'))
print({k:d.get(k) for k in ('value','ms_per_step','eager_value','eager_ms_per_step')}); print(d['config'].get('pairs_per_gpu_per_step'), d['config'].get('sub_batch_streams'), d['config'].get('latency_ms_1_pair'))
print({k:d['roofline'].get(k) for k in ('kernel','achieved','frac','avg_launch_ms','traffic')})
for r in d['rooflines']: print({k:r.get(k) for k in ('kernel','achieved','frac','avg_launch_ms','traffic')})
print(d['cpu_baseline'].get('value'), d['cpu_baseline'].get('pytorch_rocm_eager_same_gpu',{}).get('value'), d.get('other_precision',{}).get('value'))
for k,v in d.get('workloads',{}).items(): print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','eager_value','error','skipped')}, (v.get('pytorch_rocm_eager_same_gpu') or {}).get('value'))
P
