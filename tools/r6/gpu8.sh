# r6 GPU call 8: software-pipelined stride-2 marching kernel: parity, ablations, layer timing, bench A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
F='amdgpu.ids'
timeout 900 python -m pytest tests/test_gpu_march_s2.py -q -x 2>&1 | grep -v $F | tail -5
export OSA_PRECISION=f16x3
for B in 3 9; do
python tools/bench_layers.py --split --only "conv1" --batch $B --iters 30 --envs "OSA_B_RING_MASK=536879134;OSA_B_RING_MASK=8222" 2>&1 | grep -v $F | grep conv1 | tee -a $O/march_s2_layer_v2.txt
OSA_LIB_PATH=$GRAFT_REPO_ROOT/openstereo_amd/lib/variants/s2exp.so python tools/bench_layers.py --split --only "conv1" --batch $B --iters 30 --dbgs 0,1,2,3,4,7,8 2>&1 | grep -v $F | grep conv1 | tee -a $O/march_s2_ablation_v2.txt
done
unset OSA_PRECISION
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -1 | cut -c1-200 | tee $O/bench_s2v2_on_$i.json
OSA_B_RING_MASK=8222 timeout 600 python bench.py --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -1 | cut -c1-200 | tee $O/bench_s2v2_off_$i.json
done
