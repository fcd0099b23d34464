#!/bin/bash
# round 4, GPU call 4: marching kernel v2 (B operands through an LDS ring filled by LDS-DMA)
cd "$(dirname "$0")/../.."
V=openstereo_amd/lib/variants
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "marching or split_activation or pipelined or gwcnet_small or gwc_disp_processor" 2>&1 | tail -4
export OSA_PRECISION=f16x3
for B in 8 4 1; do
echo "== layers B=$B split chain"
OSA_LIB_PATH=$V/m2.so timeout 600 python tools/bench_layers.py --set 3d --batch $B --iters 10 --split --only "V0" \
   --envs "OSA_MARCH=0;OSA_MARCH_GEO=0;OSA_MARCH_GEO=1;OSA_MARCH_GEO=2;OSA_MARCH_GEO=0,OSA_MARCH_NSEG=2;OSA_MARCH_GEO=1,OSA_MARCH_NSEG=2;OSA_MARCH_GEO=1,OSA_MARCH_NSEG=4" 2>&1 | grep -v "amdgpu.ids\|redir1\|classif"
done
echo "== whole model A/B (timed only)"
OSA_LIB_PATH=$V/m2.so bash tools/bench_ab.sh "OSA_MARCH=0" "OSA_MARCH_GEO=0" "OSA_MARCH_GEO=1" "OSA_MARCH_GEO=2" 2>&1 | grep -v amdgpu.ids
