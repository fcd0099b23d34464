#!/bin/bash
# r5 call 26: the other VALU kernels that carry packed-fp32 instructions, next to the marching kernel on two other streams
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_26; mkdir -p $O; cd $R
for k in softmax upgeneric gwcvol clvol classifier head; do
  timeout 200 python tools/diag_head_under_load.py --load f16x3_split --iters 40 --kernel $k 2>&1 | grep "^\[\|iter\|Error" | head -6
done > $O/under_load.txt
cat $O/under_load.txt
