#!/bin/bash
# r5 call 24: second batch of extension ops -- ext / ctypes equality incl. the inference loops, ctypes census, model-level suites
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_24; mkdir -p $O; cd $R
timeout 500 python -m pytest tests/test_torch_ext.py -m gpu -q -s -x 2>&1 | grep -v GridwiseOp | tail -25 > $O/ext_tests.txt; cat $O/ext_tests.txt
timeout 300 python tools/ctypes_census.py 2>&1 | grep "^\[\|extension\|Error\|error" > $O/census.txt; cat $O/census.txt
timeout 600 python -m pytest tests/test_gpu_models_e2e.py tests/test_gpu_parity.py tests/test_gpu_syncbn.py tests/test_gpu_timed_config.py -m gpu -q -x 2>&1 | grep -v GridwiseOp | tail -6 > $O/models.txt; cat $O/models.txt
