cd $GRAFT_REPO_ROOT
F="grep -v amdgpu.ids\|GridwiseOp"
timeout 900 python -m pytest tests/test_feature_pyramid.py tests/test_gpu_f16_mode.py -m gpu -q 2>&1 | $F | tail -5
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models_e2e.py tests/test_gpu_timed_config.py tests/test_gpu_at_size.py -m gpu -q 2>&1 | $F | tail -4
