cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_amp_training.py tests/test_gpu_gru_train.py tests/test_gpu_autocast.py tests/test_torch_ext.py tests/test_abi_cpu.py -m gpu -q 2>&1 | grep -v GridwiseOp | tail -30 | cut -c1-300
for A in "--amp"; do python bench.py --workload stereobase_e2e_train --steps 6 --warmup 3 --no-cpu-baseline $A 2> gpurun_out/r5_e2e_amp_f16io.err | cut -c1-330; done
OSA_NATIVE_F16_IO=0 python bench.py --workload stereobase_e2e_train --steps 6 --warmup 3 --no-cpu-baseline --amp 2>/dev/null | cut -c1-330
python bench.py --workload stereobase_train --steps 10 --warmup 3 --no-cpu-baseline --amp 2>/dev/null | cut -c1-330
grep -v "GridwiseOp\|amdgpu.ids" gpurun_out/r5_e2e_amp_f16io.err | tail -5
