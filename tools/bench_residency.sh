# full launch vs taps-only (OSA_DBG 9) at 1 / 2 / 3 / 4 workgroups per CU (OSA_LDS_MIN caps residency), experiments build
export OSA_PRECISION=f16x3 OSA_LIB_PATH=openstereo_amd/lib/variants/s2u12.so
for lds in 160000 80000 53000 0; do
  echo "=== OSA_LDS_MIN=$lds"
  python tools/bench_layers.py --set 3d --batch 8 --iters 10 --dbgs 1,8,9 --env OSA_LDS_MIN=$lds --only "32->32 V0" 2>&1 | grep -v "amdgpu.ids\|sum over"
  python tools/bench_layers.py --set 3d --batch 8 --iters 10 --dbgs 1,8,9 --env OSA_LDS_MIN=$lds --only "conv2 64" 2>&1 | grep -v "amdgpu.ids\|sum over"
  python tools/bench_layers.py --set 2d --batch 8 --iters 10 --dbgs 1,8,9 --env OSA_LDS_MIN=$lds --only "l2 64" 2>&1 | grep -v "amdgpu.ids\|sum over"
done
