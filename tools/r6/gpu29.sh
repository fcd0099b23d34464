cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F="grep -v amdgpu.ids\|GridwiseOp"
timeout 900 python -m pytest tests/test_gpu_channel_sums.py -q -k "training_mode_bn or frozen" 2>&1 | $F | tail -12
timeout 900 python -m pytest tests/test_gpu_autograd.py tests/test_gpu_syncbn.py -q -x 2>&1 | $F | tail -4
( for d in 1 0 1 0; do
  echo "== OSA_TRAIN_BN=$d gwcnet_train"; OSA_TRAIN_BN=$d timeout 600 python bench.py --workload gwcnet_train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -1 | cut -c1-300
done ) | tee gpurun_out/r6/train_bn_ab.txt
