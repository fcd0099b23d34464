cd $GRAFT_REPO_ROOT
F="grep -v amdgpu.ids\|GridwiseOp"
timeout 2400 python -m pytest tests/test_gpu_syncbn.py tests/test_gpu_timed_config.py tests/test_torch_ext.py tests/test_gpu_channel_sums.py -m gpu -q 2>&1 | $F | tail -6
