cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F="grep -v amdgpu.ids\|GridwiseOp"
timeout 900 python -m pytest tests/test_gpu_autograd.py tests/test_gpu_amp_training.py -q -x 2>&1 | $F | tail -15
DEF=$(python -c "print((1<<1)|(1<<2)|(1<<3)|(1<<4)|(1<<13)|(1<<29))")
( for m in "" $DEF; do
  for b in 1 22; do
  echo "== mask=$m BATCH=$b"; OSA_B_RING_MASK=$m BATCH=$b timeout 600 python tools/bench_wgrad.py f16 f16x3 2>&1 | $F
  done
done ) | tee gpurun_out/r6/wgrad_mt_layers.txt
( for m in "" $DEF ""; do
  echo "== mask=$m amp"; OSA_B_RING_MASK=$m timeout 600 python bench.py --workload stereobase_e2e_train --amp --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -1 | cut -c1-300
done
for m in "" $DEF; do
  echo "== mask=$m f16x3"; OSA_B_RING_MASK=$m timeout 600 python bench.py --workload stereobase_e2e_train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -1 | cut -c1-300
done ) | tee gpurun_out/r6/wgrad_mt_ab.txt
