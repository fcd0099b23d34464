cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
bash tools/prof_train_graph.sh stereobase_e2e_train r6amp 210 2 --amp
cp gpurun_out/prof_r6amp/steady_state.txt gpurun_out/r6/train_amp_kernels_mt.txt
cd $GRAFT_REPO_ROOT
timeout 900 python tools/prof_train_ops.py --amp --top 150 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/train_ops_amp_mt.txt
