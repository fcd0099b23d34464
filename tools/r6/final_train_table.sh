# kernel table of the replayed StereoBase AMP training step at the final code state (tools/prof_train_graph.sh) next to the final pass
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
bash tools/prof_train_graph.sh stereobase_e2e_train r6fin 170 2 --amp
cp $GRAFT_REPO_ROOT/gpurun_out/prof_r6fin/steady_state.txt $GRAFT_REPO_ROOT/gpurun_out/r6/train_amp_kernels_final.txt
cd $GRAFT_REPO_ROOT
bash tools/profile_round6.sh
