cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | grep -v 'amdgpu.ids\|GridwiseOp' | tail -15 | tee gpurun_out/r6/suite.txt
