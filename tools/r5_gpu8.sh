# r5 GPU call 8: (a) which load kernel disturbs the fused head, (b) loads or arithmetic?  (c) pending test batches
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/openstereo_amd/lib/variants
D="python tools/diag_head_under_load.py --iters 40"
f() { grep -v amdgpu.ids | grep "^\[" | cut -c1-220; }
$D --load f16x3 --tag "march kernel (default f16x3)" 2>&1 | f
$D --load f16 --tag "brick kernel f16 + B ring" 2>&1 | f
OSA_LIB_PATH=$V/exp5.so OSA_MARCH=0 $D --load f16x3 --tag "brick f16x3 + ring (march off)" 2>&1 | f
OSA_LIB_PATH=$V/exp5.so OSA_MARCH=0 OSA_B_RING_MASK=0 OSA_DMA=0 $D --load f16x3 --tag "brick f16x3, no LDS-DMA at all" 2>&1 | f
OSA_LIB_PATH=$V/exp5.so OSA_MARCH=1 OSA_DBG=8 $D --load f16x3 --tag "march kernel without epilogue (OSA_DBG=8)" 2>&1 | f
OSA_LIB_PATH=$V/head_chk4.so $D --load f16x3 --tag "head returns checksum of its LOADS" 2>&1 | f
OSA_LIB_PATH=$V/head_chk5.so $D --load f16x3 --tag "head returns checksum of its EXPONENTIALS" 2>&1 | f
echo "=== tests"
python -m pytest tests/test_gpu_gru_train.py tests/test_gpu_syncbn.py tests/test_gpu_amp_training.py -q 2>&1 | grep -v GridwiseOp | tail -30
python -m pytest tests/test_gpu_autograd.py -q -x -k ddp 2>&1 | grep -v GridwiseOp | tail -30
python -m pytest tests/test_gpu_autograd.py tests/test_gpu_models_e2e.py tests/test_gpu_autocast.py -q --deselect tests/test_gpu_autograd.py::test_training_step_captured_as_hipgraph_under_ddp 2>&1 | grep -v GridwiseOp | tail -12
bash tools/prof_train.sh stereobase_e2e_train r5_e2e_amp_fused 300 2 --amp 2>&1 | head -64 | cut -c1-200
