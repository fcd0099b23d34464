cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 1800 python -m pytest tests/test_gpu_amp_training.py -q -s -k "whole_model" 2>&1 | grep "whole-model amp step\|passed\|failed" | tee gpurun_out/r6/whole_model_amp_pin.txt
