#!/bin/bash
# round 4, GPU call 2: timing-only ablations of the marching kernel (which resource binds the tap loop?)
cd "$(dirname "$0")/../.."
export OSA_PRECISION=f16x3
V=openstereo_amd/lib/variants
for geo in 0 1; do
for n in m_base m_nostage m_noepi m_taps m_taps_nob m_taps_statb m_taps_noa m_taps_noab m_taps_nomfma m_nob m_nomfma; do
  r=$(OSA_LIB_PATH=$V/$n.so OSA_MARCH_GEO=$geo timeout 300 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --split --only "32->32 V0" 2>&1 | grep "32->32" | sed 's/.*cfg auto://')
  echo "geo $geo $n: $r"
done
done
echo "== brick kernel reference (OSA_MARCH=0) and its ablations"
OSA_LIB_PATH=$V/m_base.so OSA_MARCH=0 timeout 300 python tools/bench_layers.py --set 3d --batch 8 --iters 10 --split --only "32->32 V0" --dbgs 1,8,9,4,13 2>&1 | grep "32->32"
