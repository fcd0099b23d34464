export OSA_PRECISION=f16x3 OSA_LIB_PATH=openstereo_amd/lib/variants/exp.so
for ks in -1 1 2 4; do
echo "== OSA_KS=$ks"
python tools/bench_layers.py --set gru --batch 4 --iters 20 --cfgs 9,11,13,14 --only gru --env OSA_KS=$ks 2>&1 | grep -v amdgpu
done
