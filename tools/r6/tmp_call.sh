cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests/test_gpu_channel_sums.py tests/test_gpu_autograd.py -q --tb=line -k "three_times or ddp_two_ranks or multi_tile" 2>&1 | grep -v "amdgpu.ids\|GridwiseOp" | grep "passed\|failed\|Error\|test_gpu_autograd.py:[0-9]" | cut -c1-300; done
