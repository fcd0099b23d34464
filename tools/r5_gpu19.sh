#!/bin/bash
# r5 call 19: gather-form lookup backward (geometry tests, StereoBase training benches) + the SyncBN spread diagnostic
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_19; mkdir -p $O; cd $R
timeout 300 python tools/diag_syncbn_spread.py > $O/syncbn_spread.txt 2>&1
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_autograd.py tests/test_torch_ext.py tests/test_gpu_models_e2e.py -m gpu -q -x > $O/tests.log 2>&1
tail -3 $O/tests.log
for a in "" "--amp"; do
  timeout 200 python bench.py --workload stereobase_e2e_train --timed-only --steps 10 --warmup 3 $a > $O/e2e_train$a.json 2> $O/e2e_train$a.err
done
OSA_GEO_BWD_SCATTER=1 timeout 200 python bench.py --workload stereobase_e2e_train --timed-only --steps 10 --warmup 3 --amp > $O/e2e_train_scatter--amp.json 2>> $O/e2e_train--amp.err
grep -h -o '"ms_per_step": [0-9.]*' $O/e2e_train*.json
cat $O/syncbn_spread.txt | tail -40
