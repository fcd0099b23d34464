#!/bin/bash
# r5 call 23: ctypes census under the extension; the rewritten FREEZE_BN test
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_23; mkdir -p $O; cd $R
timeout 400 python tools/ctypes_census.py 2>&1 | grep "^\[\|extension\|Error\|error" > $O/census.txt; cat $O/census.txt
timeout 300 python -m pytest tests/test_gpu_syncbn.py -m gpu -q -s -k freeze 2>&1 | grep -v GridwiseOp | tail -8 > $O/syncbn.txt; cat $O/syncbn.txt
