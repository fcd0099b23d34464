cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F="grep -v amdgpu.ids\|GridwiseOp"
timeout 900 python -m pytest tests/test_gpu_autograd.py -q -k "lookup" 2>&1 | $F | tail -3
for m in "--amp" "" "--amp"; do echo "== $m"; timeout 600 python bench.py --workload stereobase_e2e_train $m --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -1 | cut -c1-300; done | tee gpurun_out/r6/final_lookup_ab.txt
bash tools/prof_train_graph.sh stereobase_e2e_train r6fin 172 2 --amp
cp $GRAFT_REPO_ROOT/gpurun_out/prof_r6fin/steady_state.txt $GRAFT_REPO_ROOT/gpurun_out/r6/train_amp_kernels_final.txt
