export OSA_PRECISION=f16x3 OSA_LIB_PATH=openstereo_amd/lib/variants/w8.so
python tools/bench_layers.py --set 3d --batch 8 --iters 10 --cfgs 0,10,16 --only "V0" 2>&1 | grep -v "amdgpu.ids\|sum over"
python tools/bench_layers.py --set 3d --batch 8 --iters 10 --cfgs 1,17 --only "conv2 64" 2>&1 | grep -v "amdgpu.ids\|sum over"
