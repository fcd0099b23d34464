cd $GRAFT_REPO_ROOT
for L in none f16x3 f16 f32 gemm16; do python tools/diag_trans_probe.py --load $L 2>&1 | grep -v amdgpu.ids | cut -c1-300; done
