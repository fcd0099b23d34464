cd $GRAFT_REPO_ROOT
F="grep -v amdgpu.ids\|GridwiseOp"
timeout 1500 python -m pytest tests/test_gpu_f16_mode.py tests/test_gpu_autocast.py -q 2>&1 | $F | tail -4
for e in 1 0 1 0; do echo "== OSA_LS_F16_CHAIN=$e"; OSA_LS_F16_CHAIN=$e timeout 600 python bench.py --workload lightstereo_kitti15 --amp --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -2 | tr '\n' ' '; echo; done
