cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F="grep -v amdgpu.ids\|GridwiseOp"
( for t in cur prev cur prev; do
  if [ $t = prev ]; then cd $GRAFT_REPO_ROOT/_prev; else cd $GRAFT_REPO_ROOT; fi
  echo "== $t amp"; timeout 600 python bench.py --workload stereobase_e2e_train --amp --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -1 | cut -c1-300
done ) | tee $GRAFT_REPO_ROOT/gpurun_out/r6/prev_ab.txt
