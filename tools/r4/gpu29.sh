#!/bin/bash
# round 4, GPU call 29: full GPU suite of the final tree + whole-model sweep of the ring mask at the default line's configuration (3 x 3 pairs)
cd "$(dirname "$0")/../.."
echo "== full GPU suite"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids\|MIOpen\|c10d" | tail -4
echo "== ring mask, whole model (timed only): 0 = no ring, 8222 = default (tiles 1,2,3,4,13), 8734 = + tile 9, 30 = default without tile 13"
bash tools/bench_ab.sh "OSA_B_RING_MASK=0" "OSA_B_RING_MASK=8222" "OSA_B_RING_MASK=8734" "OSA_B_RING_MASK=30" 2>&1 | grep -v amdgpu.ids
