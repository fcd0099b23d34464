# r6 GPU call 3: single-instruction head variants + micro probes of the suspect instruction form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
F='amdgpu.ids'
echo "=== head variants, one packed instruction each"
timeout 600 python tools/diag_head_variants.py --loads march,b2,b4 --variants k_one09,k_one11,k_one20,k_one24,k_one35,packed,scalar 2>&1 | grep -v $F | tee $O/head_variants_single.txt | tail -40
echo "=== micro probes"
timeout 600 python tools/diag_pk_probe.py --victims m20,m21,m22,m23,m24,m25,m26,head_packed --loads none,march,b2,b4 2>&1 | grep -v $F | tee $O/pk_micro_matrix.txt | tail -40
