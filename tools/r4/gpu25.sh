#!/bin/bash
# round 4, GPU call 25: volume walk step 8 vs 4 at the sub-batch size of the default line (3 pairs) + torchrun launch sanity of bench.py
cd "$(dirname "$0")/../.."
echo "== volume builder, 3 pairs"
VOL_B=3 VOL_MODES=walk8split,walk4split,walk8split,walk4split,walk8,walk4 timeout 300 python tools/bench_volume.py 2>&1 | grep -v amdgpu.ids
echo "== whole model, walk step 8 vs 4"
bash tools/bench_ab.sh "OSA_VOL_WALK=8" "OSA_VOL_WALK=4" 2>&1 | grep -v amdgpu.ids
echo "== torchrun launch (1 rank)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --timed-only 2>&1 | tail -2 | cut -c1-300
