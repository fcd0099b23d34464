#!/bin/bash
# round 4, GPU call 15: final ring mask + d-walking volume builder: parity, micro-benchmark, whole-model A/B; DDP + hipGraph capture test
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "b_ring or volume" 2>&1 | tail -6
echo "== DDP + hipGraph capture (one-rank RCCL group)"
timeout 900 python -m pytest tests/test_gpu_autograd.py -q -k "captured_as_hipgraph_under_ddp" 2>&1 | tail -15
echo "== volume builder, 1 pair"
VOL_B=1 VOL_MODES=quads,walk8,walk4,quads,walk8,walk4 timeout 300 python tools/bench_volume.py 2>&1 | grep -v amdgpu.ids
echo "== volume builder, 8 pairs"
VOL_B=8 VOL_MODES=quads,walk8,walk4,quads,walk8,walk4 timeout 300 python tools/bench_volume.py 2>&1 | grep -v amdgpu.ids
echo "== whole model A/B (timed only)"
bash tools/bench_ab.sh "OSA_B_RING_MASK=0 OSA_VOL_WALK=0" "OSA_VOL_WALK=8" "OSA_VOL_WALK=4" 2>&1 | grep -v amdgpu.ids
