cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -v GridwiseOp > gpurun_out/r5_t14.log; tail -12 gpurun_out/r5_t14.log | cut -c1-300
python -m pytest tests/test_torch_ext.py -m gpu -q -s 2>&1 | grep "ctypes calls"
for A in "" "--amp"; do python bench.py --workload stereobase_e2e_train --steps 6 --warmup 3 --no-cpu-baseline $A 2>/dev/null | cut -c1-330; done
