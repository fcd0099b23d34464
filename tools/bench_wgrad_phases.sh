# wgrad_f16x3_kernel timing-only ablations (experiments build): OSA_WG_DBG bits 1 no global loads, 2 no LDS commit, 4 no MFMA phase, 8 no hand-over
export OSA_LIB_PATH=openstereo_amd/lib/variants/exp.so
for d in 0 1 2 3 4 7 8 15; do
echo "== OSA_WG_DBG=$d"; OSA_WG_DBG=$d python tools/bench_wgrad.py "3d 32->32 @48" "2d 384->128" f16x3 2>&1 | grep -v amdgpu
done
for st in 2 4 8 16; do
echo "== OSA_WGRAD_STRIP=$st"; OSA_WGRAD_STRIP=$st python tools/bench_wgrad.py "3d 32->32 @48" "2d 384->128" f16x3 2>&1 | grep -v amdgpu
done
