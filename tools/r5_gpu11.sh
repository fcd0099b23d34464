# r5 GPU call 11: whole GPU suite after the C++-extension routing + bias-in-epilogue changes; training steps
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -v GridwiseOp | tail -40 > gpurun_out/r5_t11.log; tail -30 gpurun_out/r5_t11.log
for A in "" "--amp"; do python bench.py --workload stereobase_e2e_train --steps 6 --warmup 3 --no-cpu-baseline $A 2>/dev/null | cut -c1-330; done
python bench.py --workload gwcnet_train --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-330
