#!/bin/bash
# round 4, GPU call 14: where does the B ring's time go?  timing-only ablations (experiments build) + tile sweep with the ring on / off
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
V=openstereo_amd/lib/variants
echo "== parity (ring bit-identity)"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "b_ring" 2>&1 | tail -4
export OSA_PRECISION=f16x3
for M in -1 0; do
echo "== ablations, ring mask $M: dbg 9 = no staging + no epilogue, 1024 no barrier, 2048 no transfers, 4096 no end-of-step wait"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 2d --batch 8 --iters 10 --env OSA_B_RING_MASK=$M --only "64->64" --dbgs "9,1024,2048,4096,7168,7177" 2>&1 | grep -v "amdgpu.ids\|^sum"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 2d --batch 8 --iters 10 --env OSA_B_RING_MASK=$M --only "first" --dbgs "9,1024,2048,4096,7168,7177" 2>&1 | grep -v "amdgpu.ids\|^sum"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --split --env OSA_B_RING_MASK=$M --only "conv2" --dbgs "9,1024,2048,4096,7168,7177" 2>&1 | grep -v "amdgpu.ids\|^sum"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --split --env OSA_B_RING_MASK=$M --only "conv3" --dbgs "9,1024,2048,4096,7168,7177" 2>&1 | grep -v "amdgpu.ids\|^sum"
echo "== tile sweep, ring mask $M"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 2d --batch 8 --iters 10 --env OSA_B_RING_MASK=$M --only "quarter" --cfgs "8,13,18,9,2" 2>&1 | grep -v "amdgpu.ids\|^sum"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 2d --batch 8 --iters 10 --env OSA_B_RING_MASK=$M --only "l4" --cfgs "8,13,18,9,2" 2>&1 | grep -v "amdgpu.ids\|^sum"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --split --env OSA_B_RING_MASK=$M --only "V1" --cfgs "1,4,2" 2>&1 | grep -v "amdgpu.ids\|^sum\|redir"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --split --env OSA_B_RING_MASK=$M --only "V2" --cfgs "1,4,2" 2>&1 | grep -v "amdgpu.ids\|^sum\|redir"
done
