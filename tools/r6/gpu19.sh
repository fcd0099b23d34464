cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F="grep -v amdgpu.ids\|GridwiseOp"
DEF=$(python -c "print((1<<1)|(1<<2)|(1<<3)|(1<<4)|(1<<13)|(1<<29))")
( for m in "" $DEF; do
  for b in 1 22; do
  echo "== mask=$m BATCH=$b"; OSA_B_RING_MASK=$m BATCH=$b timeout 600 python tools/bench_wgrad.py f16 f16x3 2>&1 | $F
  done
done ) | tee gpurun_out/r6/wgrad_mt_layers.txt
