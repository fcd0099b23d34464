#!/bin/bash
# round 4, GPU call 11: pixel-run depthwise kernel: parity + LightStereo workloads
cd "$(dirname "$0")/../.."
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models_e2e.py tests/test_feature_pyramid.py -q -m gpu -k "depthwise or lightstereo or pyramid" 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tail -4
for a in "" "--amp"; do
  echo "== lightstereo_kitti15 $a"; timeout 300 python bench.py --workload lightstereo_kitti15 $a --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('stage_ms_per_step'))"
  echo "== lightstereo_e2e $a"; timeout 300 python bench.py --workload lightstereo_e2e $a --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
