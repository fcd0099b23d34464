# r5 GPU call 9: head under march load -- mitigations (no packed math / no v_exp) and which march instance disturbs
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/openstereo_amd/lib/variants
D="python tools/diag_head_under_load.py --iters 40"
f() { grep -v amdgpu.ids | grep "^\[" | cut -c1-220; }
$D --load f16x3 --tag "march <0,0> fp32 in/out (76 B scratch)" 2>&1 | f
$D --load f16x3_split --tag "march <1,1> split in/out (no scratch, 250 VGPRs)" 2>&1 | f
OSA_LIB_PATH=$V/head_noslp.so $D --load f16x3 --tag "head without packed fp32 math" 2>&1 | f
OSA_LIB_PATH=$V/head_poly.so $D --load f16x3 --tag "head with polynomial exp2 (v_rcp left)" 2>&1 | f
OSA_LIB_PATH=$V/march_v200.so $D --load f16x3 --tag "march built with amdgpu_num_vgpr(200)" 2>&1 | f
python -m pytest tests/test_gpu_autograd.py -q -x -k ddp 2>&1 | grep -v GridwiseOp | tail -3
