#!/bin/bash
# round 4, GPU call 20: the ring-only 128 x 64 register tiles (configurations 16 / 17) against the tiles in use, ring on everywhere
# (run when those tiles were entries 16 / 17 of conv_cfgs.def; they live in the experiments block as 20 / 21 since -- profiles/round4/tiles_128x64_ring.txt)
cd "$(dirname "$0")/../.."
V=openstereo_amd/lib/variants
export OSA_B_RING_MASK=-1
echo "== parity vs oracle with the 128 x 64 tiles forced (3-D: 17, 2-D: 16)"
OSA_LIB_PATH=$V/exp.so OSA_CONV_CFG=17 timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "conv3d_bn_act_vs_oracle and f16x3 or split_activation_format_chain or gwc_hourglass" 2>&1 | tail -4
OSA_LIB_PATH=$V/exp.so OSA_CONV_CFG=16 timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "lightstereo_aggregation or psmnet_spp or igev_update_block" 2>&1 | tail -4
export OSA_PRECISION=f16x3
for B in 8 4; do
echo "== 2-D layers B=$B"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 2d --batch $B --iters 10 --env OSA_B_RING_MASK=-1 --only "quarter" --cfgs "13,8,16,9,13,16" 2>&1 | grep -v "amdgpu.ids\|^sum"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 2d --batch $B --iters 10 --env OSA_B_RING_MASK=-1 --only "l4" --cfgs "13,8,16,9,13,16" 2>&1 | grep -v "amdgpu.ids\|^sum"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 2d --batch $B --iters 10 --env OSA_B_RING_MASK=-1 --only "last 320" --cfgs "9,16,9,16" 2>&1 | grep -v "amdgpu.ids\|^sum"
echo "== 3-D layers B=$B (split chain)"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 3d --batch $B --iters 10 --split --env OSA_B_RING_MASK=-1 --only "V1" --cfgs "1,4,17,1,4,17" 2>&1 | grep -v "amdgpu.ids\|^sum\|redir"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set 3d --batch $B --iters 10 --split --env OSA_B_RING_MASK=-1 --only "V2" --cfgs "2,4,17,2,4,17" 2>&1 | grep -v "amdgpu.ids\|^sum\|redir"
done
echo "== gru layers B=4"
OSA_LIB_PATH=$V/exp.so timeout 600 python tools/bench_layers.py --set gru --batch 4 --iters 10 --env OSA_B_RING_MASK=-1 --only "@136x240" --cfgs "9,16,13,9,16" 2>&1 | grep -v "amdgpu.ids\|^sum"
