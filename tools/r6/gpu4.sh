# r6 GPU call 4: what about the failing form (m21: v_pk_add_f32 D, D, S op_sel:[0,1] op_sel_hi:[1,0]) matters?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
timeout 600 python tools/diag_pk_probe.py --victims m21,m30,m31,m32,m33,m34,m35,m36,m37,m38,m39 --loads none,march,b2,b4,b1,brick 2>&1 | grep -v amdgpu.ids | tee $O/pk_micro_matrix2.txt | tail -80
