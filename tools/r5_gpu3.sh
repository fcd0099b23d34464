# r5 GPU call 3: which earlier test file makes the timed-configuration test nondeterministic (it passes alone / in bench.py)?
cd $GRAFT_REPO_ROOT
export OSA_PARITY_DIAG=1
T="tests/test_gpu_timed_config.py"
run() { echo "=== $*"; timeout 600 python -m pytest "$@" -q -x -s -k "not f32 or not timed" 2>&1 | grep -E "parity diag|passed|failed|f16x3 \{" | cut -c1-900; }
run $T
run tests/test_gpu_f16x3_ranges.py $T
run tests/test_gpu_syncbn.py $T
run tests/test_gpu_parity.py $T
run tests/test_gpu_at_size.py tests/test_gpu_autocast.py tests/test_gpu_autograd.py tests/test_gpu_boundary.py $T
run tests/test_gpu_f16_mode.py tests/test_gpu_models_e2e.py $T
echo "=== with reset_arenas() first"
OSA_TEST_RESET_ARENAS=1 timeout 900 python -m pytest tests/test_gpu_f16x3_ranges.py tests/test_gpu_parity.py tests/test_gpu_syncbn.py $T -q -x -s -k "not f32 or not timed" 2>&1 | grep -E "parity diag|passed|failed|f16x3 \{" | cut -c1-900
echo "=== amax variant: gwcnet_train captured with OSA_ENGINE_AMAX=1, shipped library vs atomic-read variant"
OSA_ENGINE_AMAX=1 timeout 300 python bench.py --workload gwcnet_train --steps 4 --warmup 2 --timed-only --no-cpu-baseline 2>&1 | grep -E "capture|value" | cut -c1-300
OSA_ENGINE_AMAX=1 OSA_LIB_PATH=$GRAFT_REPO_ROOT/openstereo_amd/lib/variants/amaxld.so timeout 300 python bench.py --workload gwcnet_train --steps 4 --warmup 2 --timed-only --no-cpu-baseline 2>&1 | grep -E "capture|value" | cut -c1-300
