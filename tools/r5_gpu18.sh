cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r5_gaps; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $R/bench.py --workload stereobase_e2e_train --timed-only --steps 8 --warmup 3 --amp > $OUT/stdout.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/gap_analysis.py $F 260 22 > $OUT/gaps.txt 2>&1
rm -rf $OUT/trace
cat $OUT/gaps.txt | cut -c1-170; tail -2 $OUT/stdout.log | cut -c1-300
