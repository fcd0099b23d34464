#!/bin/bash
# Interleaved A/B of one environment switch on a training workload (GPU box, through gpurun), as used for every line of
# profiles/round6/training_steps_r6.txt:   bash tools/train_ab.sh OSA_DEFER_WGRAD [stereobase_e2e_train] [--amp]
#   switches: OSA_DEFER_WGRAD OSA_WGRAD_MULTI OSA_FROZEN_BN OSA_TRAIN_BN OSA_LOOKUP_BWD_ACC OSA_FUSED_UPSAMPLE_TRAIN OSA_FUSED_GRU_TRAIN;
#   the multi-tile weight-gradient kernels are bit 27 of OSA_B_RING_MASK (default mask 0x2800201e, bit 27 included; 0x2000201e = off)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$1; W=${2:-stereobase_e2e_train}; shift 2 2>/dev/null
for d in 1 0 1 0; do
  echo "== $V=$d $W $@"
  env $V=$d timeout 600 python bench.py --workload $W "$@" --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
