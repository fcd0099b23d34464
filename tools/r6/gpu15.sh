cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F="grep -v amdgpu.ids\|GridwiseOp"
( for d in 1 0 1; do
  echo "== OSA_DEFER_WGRAD=$d amp"; OSA_DEFER_WGRAD=$d timeout 600 python bench.py --workload stereobase_e2e_train --amp --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -3 | cut -c1-400
done
for d in 1; do
  echo "== OSA_DEFER_WGRAD=$d f16x3"; OSA_DEFER_WGRAD=$d timeout 600 python bench.py --workload stereobase_e2e_train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -3 | cut -c1-400
done ) | tee gpurun_out/r6/defer_ab.txt
timeout 2400 python -m pytest tests/test_gpu_autograd.py tests/test_gpu_amp_training.py tests/test_gpu_gru_train.py tests/test_gpu_models_e2e.py tests/test_gpu_at_size.py -q 2>&1 | $F | tail -80 | tee gpurun_out/r6/defer_tests.txt
