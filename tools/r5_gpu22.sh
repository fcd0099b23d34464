#!/bin/bash
# r5 call 22: does MIOpen's deterministic attribute remove the run-to-run spread of the GwcNet training step (torch 2-D backbone)?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_22; mkdir -p $O; cd $R
timeout 400 python tools/diag_syncbn_spread.py trace det 2>&1 | grep -v Warning > $O/trace_det.txt
grep "^run" $O/trace_det.txt | cut -c1-250
timeout 400 python tools/diag_syncbn_spread.py spread det 2>&1 | grep -v Warning | head -22 > $O/spread_det.txt
cat $O/spread_det.txt
