cd $GRAFT_REPO_ROOT
python bench.py --workload stereobase_e2e_train --steps 6 --warmup 3 --no-cpu-baseline --amp 2>/dev/null | cut -c1-330
OSA_NATIVE_F16_IO=0 python bench.py --workload stereobase_e2e_train --steps 6 --warmup 3 --no-cpu-baseline --amp 2>/dev/null | cut -c1-330
python bench.py --workload stereobase_train --steps 10 --warmup 3 --no-cpu-baseline --amp 2>/dev/null | cut -c1-330
OSA_NATIVE_F16_IO=0 python bench.py --workload stereobase_train --steps 10 --warmup 3 --no-cpu-baseline --amp 2>/dev/null | cut -c1-330
python -m pytest tests/test_gpu_amp_training.py tests/test_gpu_autograd.py -m gpu -q 2>&1 | grep -v GridwiseOp | tail -4 | cut -c1-300
bash tools/prof_train.sh stereobase_e2e_train r5_e2e_amp_f16io 300 2 --amp 2>&1 | head -40 | cut -c1-180
