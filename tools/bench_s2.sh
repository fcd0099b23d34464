export OSA_PRECISION=f16x3
for rep in 1 2; do
for v in exp s2u8 s2u12; do
  echo "=== $v"
  OSA_LIB_PATH=openstereo_amd/lib/variants/$v.so python tools/bench_layers.py --set 3d --batch 8 --iters 20 --only s2 2>&1 | grep -v "amdgpu.ids\|sum over"
done
done
