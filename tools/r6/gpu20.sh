cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F="grep -v amdgpu.ids\|GridwiseOp"
timeout 2400 python -m pytest tests/test_gpu_channel_sums.py tests/test_gpu_autograd.py tests/test_gpu_amp_training.py tests/test_gpu_gru_train.py tests/test_gpu_models_e2e.py tests/test_gpu_syncbn.py tests/test_torch_ext.py tests/test_gpu_fake_trace.py tests/test_gpu_at_size.py -q 2>&1 | $F | tail -15
