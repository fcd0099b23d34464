# classifier 32 -> 1 at 8 and 4 pairs per launch: brick form (OSA_NO_MARCH) vs the d-marching form, and the D-segment count
export OSA_PRECISION=f16x3 OSA_LIB_PATH=openstereo_amd/lib/variants/exp.so
for b in 8 4; do
echo "== batch $b: brick form"
OSA_NO_MARCH=1 python tools/bench_layers.py --batch $b --iters 30 --only classif 2>&1 | grep -v amdgpu
for ns in 0 1 2 3 4 6; do
echo "== batch $b: marching form, OSA_MARCH_NSEG=$ns (0 = automatic)"
python tools/bench_layers.py --batch $b --iters 30 --only classif --env OSA_MARCH_NSEG=$ns 2>&1 | grep -v amdgpu
done
done
