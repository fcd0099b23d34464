cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
OSA_FUSED_UPSAMPLE_TRAIN=1 bash tools/prof_train_graph.sh stereobase_e2e_train r6f1 190 2 --amp
cp gpurun_out/prof_r6f1/steady_state.txt gpurun_out/r6/train_amp_kernels_fused_up1.txt
cd $GRAFT_REPO_ROOT
OSA_FUSED_UPSAMPLE_TRAIN=0 bash tools/prof_train_graph.sh stereobase_e2e_train r6f0 185 2 --amp
cp gpurun_out/prof_r6f0/steady_state.txt gpurun_out/r6/train_amp_kernels_fused_up0.txt
