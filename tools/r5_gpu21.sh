#!/bin/bash
# r5 call 21: where in the backward pass does a deviating run of the GwcNet training step leave the others?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_21; mkdir -p $O; cd $R
timeout 400 python tools/diag_syncbn_spread.py trace 2>&1 | grep -v Warning > $O/trace.txt
OSA_PRECISION=f32 timeout 400 python tools/diag_syncbn_spread.py trace 2>&1 | grep -v Warning > $O/trace_f32.txt
tail -n 80 $O/trace.txt; tail -n 40 $O/trace_f32.txt
