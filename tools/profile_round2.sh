#!/bin/bash
# Round-2 profile (run on the GPU box through gpurun; copies of the summaries go to profiles/round2/):
#   1. rocprofv3 --kernel-trace --stats of `bench.py --timed-only --no-graph`: ONLY the timed configuration (f16x3, the bench default batch) --
#      no other-precision leg, no B=1 loop, no CPU leg -- so per-kernel averages can be read directly
#   2. the same command under --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, MI355X_MICROARCH.md): HBM bytes per launch of the
#      dominant conv instance, the volume builder, the fused head and the classifier at the bench batch -> tools/parse_pmc.py
#   3. power / clock trace (rocm-smi, 5 Hz) while the f16x3 bench, the exact-f32 bench and an 8192^3 fp16 GEMM (hipBLASLt via torch) run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r2}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --timed-only --no-graph --steps 5 --warmup 2 --batch ${BATCH:-8}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace_stdout.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- $CMD > $OUT/pmc_${C}_stdout.log 2>&1
done
python $R/tools/parse_pmc.py $OUT ${BATCH:-8} > $OUT/traffic.json 2> $OUT/parse.log
# power / clocks
smi() { while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.2; done; }
for leg in f16x3 f32 gemm; do
  smi > $OUT/smi_$leg.jsonl & SMI=$!
  if [ $leg = gemm ]; then
    timeout 120 python - > $OUT/gemm.log 2>&1 <<PY
import torch, time
a = torch.randn(8192, 8192, device="cuda", dtype=torch.float16); b = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
for _ in range(20): a @ b
torch.cuda.synchronize(); t = time.perf_counter(); n = 400
for _ in range(n): a @ b
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print(f"fp16 GEMM 8192^3 (torch.matmul -> hipBLASLt), uniform-random operands: {dt*1e3:.3f} ms = {2*8192**3/dt/1e12:.1f} TFLOP/s")
PY
  else
    timeout 200 python $R/bench.py --timed-only --precision $leg --steps $([ $leg = f32 ] && echo 40 || echo 80) --warmup 5 --batch ${BATCH:-8} > $OUT/bench_$leg.json 2>/dev/null
  fi
  kill $SMI; wait $SMI 2>/dev/null
done
python $R/tools/parse_smi.py $OUT > $OUT/power_summary.txt 2>&1
ls $OUT
