cd $GRAFT_REPO_ROOT
F="grep -v amdgpu.ids\|GridwiseOp"
for e in "" 1 "" 1; do echo "== OSA_EXP_POOL_CL=$e"; OSA_EXP_POOL_CL=$e timeout 600 python bench.py --workload stereobase_e2e_train --amp --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | $F | tail -1 | cut -c1-300; done
