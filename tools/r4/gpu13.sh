#!/bin/bash
# round 4, GPU call 13: B operands through the LDS ring in the brick kernel (conv_kernel.h BL = 1): parity, per-layer A/B, whole-model A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
echo "== parity (ring bit-identity + conv suites)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "b_ring or conv3d_bn or deconv3d or conv2d or gwcnet_small or gwc_disp_processor or psmnet_256 or split_activation or lightstereo_aggregation or igev_update" 2>&1 | tail -8
export OSA_PRECISION=f16x3
echo "== 3-D layers B=8 (split chain)"
timeout 600 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --split --envs "OSA_B_RING_MASK=0;OSA_B_RING_MASK=-1" 2>&1 | grep -v "amdgpu.ids"
echo "== 3-D layers B=4 (split chain)"
timeout 600 python tools/bench_layers.py --set 3d --batch 4 --iters 10 --split --envs "OSA_B_RING_MASK=0;OSA_B_RING_MASK=-1" 2>&1 | grep -v "amdgpu.ids"
echo "== 2-D layers B=8"
timeout 600 python tools/bench_layers.py --set 2d --batch 8 --iters 10 --envs "OSA_B_RING_MASK=0;OSA_B_RING_MASK=-1" 2>&1 | grep -v "amdgpu.ids"
echo "== 2-D layers B=4"
timeout 600 python tools/bench_layers.py --set 2d --batch 4 --iters 10 --envs "OSA_B_RING_MASK=0;OSA_B_RING_MASK=-1" 2>&1 | grep -v "amdgpu.ids"
echo "== gru layers B=4"
timeout 600 python tools/bench_layers.py --set gru --batch 4 --iters 10 --envs "OSA_B_RING_MASK=0;OSA_B_RING_MASK=-1" 2>&1 | grep -v "amdgpu.ids"
echo "== whole model A/B (timed only)"
bash tools/bench_ab.sh "OSA_B_RING_MASK=0" "OSA_B_RING_MASK=-1" 2>&1 | grep -v amdgpu.ids
unset OSA_PRECISION
echo "== amax kernel (slots cleared by atomics) in the captured GwcNet training step"
OSA_ENGINE_AMAX=1 timeout 600 python bench.py --workload gwcnet_train --steps 10 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-400
