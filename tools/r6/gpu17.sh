cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F="grep -v amdgpu.ids\|GridwiseOp"
( for d in 1 0 1; do
  echo "== OSA_SIDE_WGRAD=$d amp"; OSA_SIDE_WGRAD=$d timeout 600 python bench.py --workload stereobase_e2e_train --amp --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -3 | cut -c1-400
done
for d in 1; do
  echo "== OSA_SIDE_WGRAD=$d f16x3"; OSA_SIDE_WGRAD=$d timeout 600 python bench.py --workload stereobase_e2e_train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -3 | cut -c1-400
done ) | tee gpurun_out/r6/side_ab.txt
