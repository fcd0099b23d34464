#!/bin/bash
# r5 call 25: GwcNet training step, 2-D extractor through MIOpen (default) vs through the engine's differentiable convolutions
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_25; mkdir -p $O; cd $R
for e in 0 1; do
  OSA_GWC_TRAIN_BACKBONE_ENGINE=$e timeout 300 python bench.py --workload gwcnet_train --timed-only --steps 10 --warmup 3 > $O/gwcnet_train_e$e.json 2> $O/gwcnet_train_e$e.err
  grep -o '"ms_per_step": [0-9.]*' $O/gwcnet_train_e$e.json
done
OSA_GWC_TRAIN_BACKBONE_ENGINE=1 timeout 300 python -m pytest tests/test_gpu_autograd.py tests/test_gpu_models_e2e.py -m gpu -q -x -k "gwc or Gwc or training" 2>&1 | grep -v GridwiseOp | tail -4
