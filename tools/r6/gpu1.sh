# r6 GPU call 1: packed-fp32 trigger matrix, amax re-test with the packed-free library, widened concurrency test, A/B of the build flags
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
V=$GRAFT_REPO_ROOT/openstereo_amd/lib/variants
F='amdgpu.ids'
echo "=== pk probe matrix" 
timeout 900 python tools/diag_pk_probe.py 2>&1 | grep -v $F | tee $O/pk_probe_matrix.txt | tail -200
echo "=== amax: gwcnet_train captured with OSA_ENGINE_AMAX=1: packed-free library, then the r5-flags library"
(OSA_ENGINE_AMAX=1 timeout 300 python bench.py --workload gwcnet_train --steps 4 --warmup 2 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -3 | cut -c1-600) 2>&1 | tee $O/amax_nopk.txt
(OSA_ENGINE_AMAX=1 OSA_LIB_PATH=$V/r5flags.so timeout 300 python bench.py --workload gwcnet_train --steps 4 --warmup 2 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -3 | cut -c1-600) 2>&1 | tee $O/amax_r5flags.txt
echo "=== concurrency + timed-config tests"
timeout 1500 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_timed_config.py -q -x 2>&1 | grep -v $F | tail -15 | tee $O/concurrency_tests.txt
echo "=== bench A/B: packed-free (shipped) vs r5 flags"
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -1 | cut -c1-1500 | tee $O/bench_nopk_$i.json
OSA_LIB_PATH=$V/r5flags.so timeout 600 python bench.py --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -1 | cut -c1-1500 | tee $O/bench_r5flags_$i.json
done
