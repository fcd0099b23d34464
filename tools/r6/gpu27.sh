cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F="grep -v amdgpu.ids\|GridwiseOp"
echo "== alone"; timeout 600 python -m pytest tests/test_gpu_f16_mode.py -q 2>&1 | $F | tail -4
echo "== after channel_sums"; timeout 600 python -m pytest tests/test_gpu_channel_sums.py tests/test_gpu_f16_mode.py -q 2>&1 | $F | tail -4
echo "== after concurrency"; timeout 900 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_f16_mode.py -q 2>&1 | $F | tail -4
echo "== rest of the suite"; timeout 3000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_f16_mode.py::test_autocast_region_selects_the_f16_mode 2>&1 | $F | tail -8 | tee gpurun_out/r6/suite.txt
