#!/bin/bash
# round 4, GPU call 23: walk kernel with incremental address arithmetic: parity + micro-benchmark (fp32 and split output) + DDP test
cd "$(dirname "$0")/../.."
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "volume or gwcnet_small" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_autograd.py -q -k "ddp_two_ranks" 2>&1 | tail -3
echo "== volume builder, 8 pairs"
VOL_B=8 VOL_MODES=quads,walk8,walk8split,quads,walk8,walk8split timeout 300 python tools/bench_volume.py 2>&1 | grep -v amdgpu.ids
echo "== volume builder, 3 pairs"
VOL_B=3 VOL_MODES=quads,walk8,walk8split,quads,walk8,walk8split timeout 300 python tools/bench_volume.py 2>&1 | grep -v amdgpu.ids
