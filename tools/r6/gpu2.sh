# r6 GPU call 2: delta-debugged head variants x loads; refined burners; whole-model concurrency tests under MIOpen's deterministic attribute
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
F='amdgpu.ids'
echo "=== head variants"
timeout 900 python tools/diag_head_variants.py --loads none,march,b2,b4 2>&1 | grep -v $F | tee $O/head_variants_matrix.txt | tail -150
echo "=== refined burners vs the packed head"
timeout 600 python tools/diag_pk_probe.py --victims head_packed,head,p5,p6 --loads b1,b7,b8,b9,b10,b4,b2 2>&1 | grep -v $F | tee $O/pk_probe_burners2.txt | tail -40
echo "=== whole-model concurrency tests"
timeout 900 python -m pytest tests/test_gpu_concurrency.py -q -x -k whole_forwards 2>&1 | grep -v $F | tail -8 | tee $O/concurrency_models.txt
