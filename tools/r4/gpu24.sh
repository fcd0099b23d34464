#!/bin/bash
# round 4, GPU call 24: profile passes of the final code (split volume, tile 4 under the ring) + full GPU suite
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
echo "== profile"
SKIP_SQ= bash tools/profile_round4.sh r4c > gpurun_out/r4/profile_c.log 2>&1; tail -2 gpurun_out/r4/profile_c.log; head -c 900 gpurun_out/prof_r4c/traffic.json; echo; cat gpurun_out/prof_r4c/sq_summary.txt 2>/dev/null | tail -9
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
echo "== full GPU suite"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids\|MIOpen\|c10d" | tail -8
