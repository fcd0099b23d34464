# r5 GPU call 4: locate the replay-to-replay nondeterminism of the timed configuration (f16x3, 3 sub-batch streams, hipGraph)
cd $GRAFT_REPO_ROOT
D="python tools/diag_timed_config.py"
V=$GRAFT_REPO_ROOT/openstereo_amd/lib/variants
for i in 1 2; do $D --tag default_$i 2>&1 | grep -v amdgpu.ids; done
$D --streams 1 --tag one_stream_graph 2>&1 | grep -v amdgpu.ids
$D --no-graph --tag three_streams_eager 2>&1 | grep -v amdgpu.ids
$D --precision f32 --tag f32 2>&1 | grep -v amdgpu.ids
$D --warm 30 --tag warm30 2>&1 | grep -v amdgpu.ids
OSA_B_RING_MASK=0 $D --tag no_b_ring 2>&1 | grep -v amdgpu.ids
OSA_VOL_WALK=0 $D --tag chunked_volume 2>&1 | grep -v amdgpu.ids
OSA_VOL_SPLIT=0 $D --tag fp32_volume 2>&1 | grep -v amdgpu.ids
OSA_SPLIT_ACT=0 $D --tag no_split_activations 2>&1 | grep -v amdgpu.ids
OSA_LIB_PATH=$V/exp5.so OSA_MARCH=0 $D --tag exp_no_march 2>&1 | grep -v amdgpu.ids
OSA_LIB_PATH=$V/exp5.so OSA_DMA=0 $D --tag exp_no_dma_staging 2>&1 | grep -v amdgpu.ids
OSA_LIB_PATH=$V/exp5.so OSA_MARCH=0 OSA_DMA=0 OSA_B_RING_MASK=0 OSA_VOL_WALK=0 $D --tag exp_no_lds_dma_at_all 2>&1 | grep -v amdgpu.ids
OSA_LIB_PATH=$V/amaxld.so $D --tag amax_atomic_reads 2>&1 | grep -v amdgpu.ids
echo "=== syncbn gwcnet freeze test"
python -m pytest tests/test_gpu_syncbn.py -q -x -s -k freeze 2>&1 | tail -15
echo "=== amax: gwcnet_train captured with OSA_ENGINE_AMAX=1, shipped library vs atomic-read variant"
OSA_ENGINE_AMAX=1 timeout 300 python bench.py --workload gwcnet_train --steps 4 --warmup 2 --timed-only --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400
OSA_ENGINE_AMAX=1 OSA_LIB_PATH=$V/amaxld.so timeout 300 python bench.py --workload gwcnet_train --steps 4 --warmup 2 --timed-only --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400
