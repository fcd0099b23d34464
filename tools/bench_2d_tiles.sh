# 2-D backbone layers at 8 pairs (16 images) per launch: register-tile configurations (experiments build)
export OSA_PRECISION=f16x3 OSA_LIB_PATH=openstereo_amd/lib/variants/exp.so
python tools/bench_layers.py --set 2d --batch 8 --iters 20 --cfgs 7,8,9,12,13,18,19,20,21,22 2>&1 | grep -v amdgpu
