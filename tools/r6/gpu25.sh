cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F="grep -v amdgpu.ids\|GridwiseOp"
timeout 2400 python -m pytest tests/test_gpu_channel_sums.py tests/test_gpu_models_e2e.py tests/test_gpu_amp_training.py tests/test_torch_ext.py tests/test_gpu_fake_trace.py -q -x 2>&1 | $F | tail -8
( for d in 1 0 1; do
  echo "== OSA_FUSED_UPSAMPLE_TRAIN=$d amp"; OSA_FUSED_UPSAMPLE_TRAIN=$d timeout 600 python bench.py --workload stereobase_e2e_train --amp --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -1 | cut -c1-300
done
echo "== f16x3";  timeout 600 python bench.py --workload stereobase_e2e_train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -1 | cut -c1-300
) | tee gpurun_out/r6/upsample_ab.txt
