# r6 GPU call 7: timing-only ablations of the stride-2 marching kernel (experiments build): what binds it?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
export OSA_PRECISION=f16x3 OSA_LIB_PATH=$GRAFT_REPO_ROOT/openstereo_amd/lib/variants/s2exp.so
for B in 3 9; do
python tools/bench_layers.py --split --only "conv1" --batch $B --iters 30 --dbgs 0,1,2,3,4,5,6,7,8 2>&1 | grep -v amdgpu.ids | grep conv1 | tee -a $O/march_s2_ablation.txt
done
