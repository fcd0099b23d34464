cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_autograd.py -q -k "ddp_two_ranks" 2>&1 | grep -v "amdgpu.ids\|GridwiseOp" | grep "passed\|failed\|assert \|Error\|^E " | tail -6; done
