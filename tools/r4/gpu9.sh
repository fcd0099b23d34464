#!/bin/bash
# round 4, GPU call 9: whole GPU suite (no -x) + round-4 profile passes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
echo "== full GPU suite"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tail -40
echo "== profile"
SKIP_SQ= bash tools/profile_round4.sh r4 > gpurun_out/r4/profile.log 2>&1; tail -3 gpurun_out/r4/profile.log; cat gpurun_out/prof_r4/traffic.json | head -30
