cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 900 python bench.py --workload stereobase_e2e_train --amp --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r6/e2e_train_amp_with_eager.json
