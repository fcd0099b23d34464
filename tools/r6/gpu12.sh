cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fake_trace.py tests/test_torch_ext.py -q -x 2>&1 | grep -v 'amdgpu.ids' | tail -40
