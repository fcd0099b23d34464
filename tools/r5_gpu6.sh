# r5 GPU call 6: (a) is the fused head deterministic under concurrent MFMA load?  (b) fused GRU training path: tests + AMP step time
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/openstereo_amd/lib/variants
D="python tools/diag_head_under_load.py"
f() { grep -v amdgpu.ids | cut -c1-220; }
$D --load none --tag head_idle 2>&1 | f
$D --load f16x3 --tag head_f16x3_load 2>&1 | f
$D --load f16x3 --tag head_f16x3_load_again 2>&1 | f
$D --load f32 --tag head_f32_load 2>&1 | f
$D --load f16x3 --kernel copy --tag copy_f16x3_load 2>&1 | f
$D --load f16x3 --kernel classifier --tag classifier_f16x3_load 2>&1 | f
OSA_LIB_PATH=$V/head_sc1.so $D --load f16x3 --tag head_sc1_loads 2>&1 | f
OSA_LIB_PATH=$V/head_inv.so $D --load f16x3 --tag head_l1_invalidate 2>&1 | f
echo "=== tests"
python -m pytest tests/test_gpu_gru_train.py tests/test_gpu_syncbn.py tests/test_gpu_amp_training.py -q 2>&1 | tail -25
python -m pytest tests/test_gpu_autograd.py tests/test_gpu_models_e2e.py tests/test_gpu_autocast.py -q -x 2>&1 | tail -8
echo "=== AMP whole-model training step with the fused GRU path (before: 147.0 ms)"
python bench.py --workload stereobase_e2e_train --steps 6 --warmup 3 --no-cpu-baseline --amp 2> gpurun_out/r5_e2e_amp_fused.err | cut -c1-400
OSA_FUSED_GRU_TRAIN=0 python bench.py --workload stereobase_e2e_train --steps 6 --warmup 3 --no-cpu-baseline --amp 2>/dev/null | cut -c1-300
python bench.py --workload stereobase_e2e_train --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-300
tail -3 gpurun_out/r5_e2e_amp_fused.err
