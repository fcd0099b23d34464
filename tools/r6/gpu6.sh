# r6 GPU call 6: the stride-2 marching kernel as a single layer -- timing at 3 and 9 pairs vs the brick form, counters at 9 pairs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
F='amdgpu.ids'
export OSA_PRECISION=f16x3
for B in 3 9; do
python tools/bench_layers.py --split --only "conv1" --batch $B --iters 30 --envs "OSA_B_RING_MASK=536879134;OSA_B_RING_MASK=8222" 2>&1 | grep -v $F | grep conv1 | tee -a $O/march_s2_layer.txt
done
bash tools/r6/pmc_layer.sh s2_on -- env OSA_B_RING_MASK=536879134 python $GRAFT_REPO_ROOT/tools/bench_layers.py --split --only conv1 --batch 9 --iters 10 2>&1 | grep -v $F | tee $O/march_s2_pmc_on.txt
bash tools/r6/pmc_layer.sh s2_off -- env OSA_B_RING_MASK=8222 python $GRAFT_REPO_ROOT/tools/bench_layers.py --split --only conv1 --batch 9 --iters 10 2>&1 | grep -v $F | tee $O/march_s2_pmc_off.txt
