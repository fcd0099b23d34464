#!/bin/bash
# r5 call 27: three forms of the lookup gradient timed; tests + training bench with the rows form
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_27; mkdir -p $O; cd $R
OSA_LIB_PATH=openstereo_amd/lib/variants/exp_geo.so timeout 200 python tools/bench_lookup_bwd.py 2>&1 | grep "^\[\|Error" > $O/lookup_bwd_forms.txt; cat $O/lookup_bwd_forms.txt
timeout 400 python -m pytest tests/test_gpu_autograd.py tests/test_torch_ext.py tests/test_gpu_autocast.py -m gpu -q -x 2>&1 | grep -v GridwiseOp | tail -3
timeout 200 python bench.py --workload stereobase_e2e_train --timed-only --steps 10 --warmup 3 --amp > $O/e2e_train_amp.json 2> $O/e2e_train_amp.err; grep -o '"ms_per_step": [0-9.]*' $O/e2e_train_amp.json
