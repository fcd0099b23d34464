# r6 GPU call 13: fused transposed conv as 8-wave workgroups (128 positions x 64 channels): parity, layer timing, bench A/B   (mask bit 27)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
F='amdgpu.ids'
M_ON=671096862   # default 536879134 + 2^27
OSA_B_RING_MASK=$M_ON timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "deconv or gwcnet or hourglass or redir" 2>&1 | grep -v $F | tail -6
export OSA_PRECISION=f16x3
for B in 3 9; do
python tools/bench_layers.py --only "conv5" --batch $B --iters 30 --envs "OSA_B_RING_MASK=536879134;OSA_B_RING_MASK=$M_ON" 2>&1 | grep -v $F | grep conv5 | tee -a $O/deconv_w8_layer.txt
done
unset OSA_PRECISION
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -1 | cut -c1-200 | tee $O/bench_dw8_off_$i.json
OSA_B_RING_MASK=$M_ON timeout 600 python bench.py --steps 20 --warmup 5 --timed-only --no-cpu-baseline 2>&1 | grep -v $F | tail -1 | cut -c1-200 | tee $O/bench_dw8_on_$i.json
done
