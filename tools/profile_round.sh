#!/bin/bash
# Round profile: kernel-trace stats of the default bench + HBM traffic counters (separate PMC passes, as
# MI355X_MICROARCH.md prescribes) for the dominant conv launch and the volume builder.
# Usage (on the GPU box):  bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-round}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $R/bench.py --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  OSA_PRECISION=f16x3 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${C}_conv -o p -- python $R/tools/bench_layers.py --only "32->32 V0" --iters 3 > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${C}_vol -o p -- python $R/tools/bench_volume.py > /dev/null 2>&1
done
find $OUT -name "*.csv" | head -30
