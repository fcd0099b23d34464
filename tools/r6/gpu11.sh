# r6 GPU call 11: stride-2 marching kernel, 4-wave (two workgroups per CU) vs 8-wave (one) vs brick
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
F='amdgpu.ids'
timeout 900 python -m pytest tests/test_gpu_march_s2.py -q -x 2>&1 | grep -v $F | tail -5
OSA_B_RING_MASK=805314590 timeout 900 python -m pytest tests/test_gpu_march_s2.py -q -x -k "torch_and_brick" 2>&1 | grep -v $F | tail -3
export OSA_PRECISION=f16x3
for B in 3 9; do
python tools/bench_layers.py --split --only "conv1" --batch $B --iters 30 --envs "OSA_B_RING_MASK=536879134;OSA_B_RING_MASK=805314590;OSA_B_RING_MASK=8222" 2>&1 | grep -v $F | grep conv1 | tee -a $O/march_s2_layer_w4.txt
done
bash tools/r6/pmc_layer.sh s2_w4 -- env OSA_B_RING_MASK=536879134 python $GRAFT_REPO_ROOT/tools/bench_layers.py --split --only conv1 --batch 9 --iters 10 2>&1 | grep -v $F | grep 'conv_march_s2' | tee $O/march_s2_pmc_w4.txt
unset OSA_PRECISION
cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -1 | cut -c1-200 | tee $O/bench_s2w4_$i.json
OSA_B_RING_MASK=805314590 timeout 600 python bench.py --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -1 | cut -c1-200 | tee $O/bench_s2w8_$i.json
OSA_B_RING_MASK=8222 timeout 600 python bench.py --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -1 | cut -c1-200 | tee $O/bench_s2off_$i.json
done
