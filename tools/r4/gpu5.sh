#!/bin/bash
# round 4, GPU call 5: marching kernel v3 (4-slot B ring, 2-step latency tolerance) + its ablations
cd "$(dirname "$0")/../.."
V=openstereo_amd/lib/variants
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "marching or split_activation or pipelined or gwcnet_small or gwc_disp_processor" 2>&1 | tail -4
export OSA_PRECISION=f16x3
for geo in 1 2; do
for n in m3 m3_nobar m3_nowait m3_nodma m3_nobread m3_nostage m3_noepi m3_taps m3_taps_nobar m3_taps_nob; do
  r=$(OSA_LIB_PATH=$V/$n.so OSA_MARCH_GEO=$geo timeout 300 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --split --only "32->32 V0" 2>&1 | grep "32->32" | sed 's/.*cfg auto://')
  echo "geo $geo $n: $r"
done
done
for B in 8 4; do
echo "== layers B=$B split chain"
OSA_LIB_PATH=$V/m3.so timeout 600 python tools/bench_layers.py --set 3d --batch $B --iters 10 --split --only "V0" \
   --envs "OSA_MARCH=0;OSA_MARCH_GEO=0;OSA_MARCH_GEO=1;OSA_MARCH_GEO=2;OSA_MARCH_GEO=1,OSA_MARCH_NSEG=2" 2>&1 | grep -v "amdgpu.ids\|redir1\|classif"
done
echo "== whole model A/B (timed only)"
OSA_LIB_PATH=$V/m3.so bash tools/bench_ab.sh "OSA_MARCH=0" "OSA_MARCH_GEO=1" "OSA_MARCH_GEO=2" 2>&1 | grep -v amdgpu.ids
