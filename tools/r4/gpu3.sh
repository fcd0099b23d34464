#!/bin/bash
# round 4, GPU call 3: marching ablations + f16-mode parity + refactor regression (conv parity subset)
cd "$(dirname "$0")/../.."
V=openstereo_amd/lib/variants
echo "== f16 mode tests"; timeout 900 python -m pytest tests/test_gpu_f16_mode.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25
echo "== refactor regression (conv parity, both modes)"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv3d_bn_act or deconv3d or marching or split_activation or gwcnet_small or igev_update or lightstereo_aggregation_vs or deconv2d" 2>&1 | tail -4
export OSA_PRECISION=f16x3
for geo in 0 1; do
for n in m_base m_nostage m_noepi m_taps m_taps_nob m_taps_statb m_taps_noa m_taps_noab m_taps_nomfma m_nob m_nomfma; do
  r=$(OSA_LIB_PATH=$V/$n.so OSA_MARCH_GEO=$geo timeout 300 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --split --only "32->32 V0" 2>&1 | grep "32->32" | sed 's/.*cfg auto://')
  echo "geo $geo $n: $r"
done
done
echo "== brick kernel reference (OSA_MARCH=0) and its ablations"
OSA_LIB_PATH=$V/m_base.so OSA_MARCH=0 timeout 300 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --split --only "32->32 V0" --dbgs 1,8,9,4,13 2>&1 | grep "32->32"
