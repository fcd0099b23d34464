#!/bin/bash
# round 4, GPU call 21: the volume written as a split tensor (dres0.0 stages it by LDS-DMA), tile 4 under the ring: parity + whole-model A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "volume or b_ring or gwcnet or split_activation or gwc_disp_processor or gwc_hourglass or marching" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_torch_ext.py tests/test_gpu_at_size.py -q -m gpu 2>&1 | tail -3
echo "== whole model A/B (timed only)"
bash tools/bench_ab.sh "OSA_VOL_SPLIT=0" "OSA_VOL_SPLIT=1" 2>&1 | grep -v amdgpu.ids
