cd $GRAFT_REPO_ROOT
python bench.py --workload stereobase_e2e_train --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r5_e2e_f16x3.json 2> gpurun_out/r5_e2e_f16x3.err; tail -c 1500 gpurun_out/r5_e2e_f16x3.err | grep -v GridwiseOp; head -c 300 gpurun_out/r5_e2e_f16x3.json
echo "=== f16_mode alone"
python -m pytest tests/test_gpu_f16_mode.py -q -x 2>&1 | grep -v GridwiseOp | tail -5
echo "=== suite in order, verbose names"
python -m pytest tests -m gpu -q -x -v 2>&1 | grep -v GridwiseOp | grep -E "PASSED|FAILED|ERROR|Fatal|fault|passed|failed" | tail -25 | cut -c1-200
