# r6 GPU call 9: SQ counters of the stride-2 marching kernel (shipped build) at 9 pairs
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6/sq_s2; mkdir -p $O
export OSA_PRECISION=f16x3
CMD="python $GRAFT_REPO_ROOT/tools/bench_layers.py --split --only conv1 --batch 9 --iters 5"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o 'SQ_[A-Z_0-9]*' | sort -u | tr '\n' ' ' > $O/sq_counters.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -o p -- $CMD > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $O/p2 -o p -- $CMD > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/p3 -o p -- $CMD > $O/p3.log 2>&1
python - $O <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_march_s2" in r["Kernel_Name"] or "conv_mfma_kernel<1, 1, 3" in r["Kernel_Name"]:
            acc.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]; print(f"{k[0]:40s} {k[1]:28s} avg {sum(v)/len(v):14.4g} over {len(v)} dispatches")
PY
tail -3 $O/p2.log $O/p3.log | grep -i 'error\|invalid\|not' | head
