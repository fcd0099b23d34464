# r6 GPU call 5: stride-2 marching kernel -- parity, then A/B on the default bench line and per-layer timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
F='amdgpu.ids'
timeout 900 python -m pytest tests/test_gpu_march_s2.py -q -x 2>&1 | grep -v $F | tail -25 | tee $O/march_s2_tests.txt
echo "=== GwcNet goldens + timed config"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_config.py -q -x -k "gwcnet or timed or psmnet" 2>&1 | grep -v $F | tail -8 | tee $O/march_s2_goldens.txt
echo "=== bench A/B (bit 29 on / off)"
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -1 | cut -c1-300 | tee $O/bench_s2_on_$i.json
OSA_B_RING_MASK=8222 timeout 600 python bench.py --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -1 | cut -c1-300 | tee $O/bench_s2_off_$i.json
done
