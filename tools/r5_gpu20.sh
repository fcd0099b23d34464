#!/bin/bash
# r5 call 20: is the run-to-run spread of the GwcNet training step a read of unwritten memory?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_20; mkdir -p $O; cd $R
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 400 python tools/diag_syncbn_spread.py 2>&1 | grep -v Warning | head -24 > $O/spread_nocache.txt
OSA_TORCH_EXT=0 timeout 300 python tools/diag_syncbn_spread.py poison=nan 2>&1 | grep -v Warning > $O/poison_nan.txt
OSA_TORCH_EXT=0 timeout 300 python tools/diag_syncbn_spread.py poison=30000 2>&1 | grep -v Warning > $O/poison_3e4.txt
OSA_TORCH_EXT=0 timeout 300 python tools/diag_syncbn_spread.py 2>&1 | grep -v Warning | head -24 > $O/spread_noext.txt
tail -n 30 $O/*.txt
