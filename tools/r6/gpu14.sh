cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 900 python tools/prof_train_ops.py --amp 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/train_ops_amp.txt | tail -120
