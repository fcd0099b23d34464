#!/bin/bash
# round 4, GPU call 16: XCD-aware head tiles (parity + timing), DDP capture test, default bench line without the extra workloads
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
echo "== parity (heads, whole GwcNet)"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "regression or disp_processor or gwcnet_small or gwcnet_full_size" 2>&1 | tail -4
echo "== DDP + hipGraph capture (one-rank RCCL group)"
timeout 900 python -m pytest tests/test_gpu_autograd.py -q -k "captured_as_hipgraph_under_ddp" 2>&1 | tail -25
echo "== default bench, no extra workloads"
timeout 1200 python bench.py --no-workloads 2>gpurun_out/r4/bench16.err | tail -1 > gpurun_out/r4/bench16.json; python - <<'P'
import json
d=json.load(open('gpurun_out/r4/bench16.json'))
print({k:d.get(k) for k in ('value','ms_per_step','eager_value','eager_ms_per_step')}); print(d['roofline'])
for r in d['rooflines']: print({k:r.get(k) for k in ('kernel','achieved','frac','avg_launch_ms','traffic')})
print(d['config'].get('stage_ms_per_step'))
print(d['cpu_baseline'].get('value'), d['cpu_baseline'].get('pytorch_rocm_eager_same_gpu',{}).get('value'))
P
tail -5 gpurun_out/r4/bench16.err
