# r5 GPU call 5: which stage of the forward differs first between concurrent sub-batch streams and the one-stream run?
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/openstereo_amd/lib/variants
python tools/diag_timed_config.py --no-graph --stages --replays 5 2>&1 | grep -v amdgpu.ids | cut -c1-260
echo "=== march off (experiments build)"
OSA_LIB_PATH=$V/exp5.so OSA_MARCH=0 python tools/diag_timed_config.py --no-graph --stages --replays 4 2>&1 | grep -v amdgpu.ids | cut -c1-260
for i in 1 2 3; do OSA_LIB_PATH=$V/exp5.so OSA_MARCH=0 python tools/diag_timed_config.py --tag exp_no_march_$i 2>&1 | grep -v amdgpu.ids | tail -1; done
for i in 1 2 3; do python tools/diag_timed_config.py --precision f32 --tag f32_$i 2>&1 | grep -v amdgpu.ids | tail -1; done
