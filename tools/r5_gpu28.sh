#!/bin/bash
# r5 call 28: row-parallel forward lookup of the training path: forms timed, tests, training benches
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_28; mkdir -p $O; cd $R
OSA_LIB_PATH=openstereo_amd/lib/variants/exp_geo.so timeout 200 python tools/bench_lookup.py 2>&1 | grep "^\[\|Error" > $O/lookup_forms.txt; cat $O/lookup_forms.txt
timeout 500 python -m pytest tests/test_gpu_autograd.py tests/test_torch_ext.py tests/test_gpu_parity.py tests/test_gpu_models_e2e.py -m gpu -q -x 2>&1 | grep -v GridwiseOp | tail -3
for a in "--amp" ""; do timeout 200 python bench.py --workload stereobase_e2e_train --timed-only --steps 10 --warmup 3 $a > $O/e2e_train$a.json 2> $O/e2e_train$a.err; grep -o '"ms_per_step": [0-9.]*' $O/e2e_train$a.json; done
