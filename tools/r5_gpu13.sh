cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -v GridwiseOp > gpurun_out/r5_t13.log; tail -15 gpurun_out/r5_t13.log | cut -c1-300
for A in "" "--amp"; do python bench.py --workload stereobase_e2e_train --steps 6 --warmup 3 --no-cpu-baseline $A 2>/dev/null | cut -c1-330; done
