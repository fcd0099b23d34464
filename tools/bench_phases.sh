# timing-only ablations (experiments build, OSA_DBG): 1 = no staging, 8 = no epilogue, 9 = taps only, 32 = no stores
export OSA_PRECISION=f16x3 OSA_LIB_PATH=openstereo_amd/lib/variants/s2u12.so
python tools/bench_layers.py --set 3d --batch 8 --iters 10 --dbgs 1,8,9,32 2>&1 | grep -v "amdgpu.ids"
python tools/bench_layers.py --set 2d --batch 8 --iters 10 --dbgs 1,8,9,32 2>&1 | grep -v "amdgpu.ids"
