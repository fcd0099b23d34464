#!/bin/bash
# round 4, GPU call 10: kernel tables of the AMP workloads (where does the f16-mode IGEV loop / LightStereo stage spend its time?) + ext test
cd "$(dirname "$0")/../.."
R=$(pwd)
mkdir -p gpurun_out/r4
echo "== torch extension vs ctypes"; timeout 600 python -m pytest tests/test_torch_ext.py tests/test_gpu_autograd.py -q -m gpu -k "extension or ddp" 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tail -4
cd /tmp && export TMPDIR=/tmp
for wl in igev_refine32 lightstereo_kitti15 stereobase_e2e; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof_${wl}_amp -o p -- python $R/bench.py --workload $wl --amp --no-graph --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r4/prof_${wl}_amp.log 2>&1
  f=$(find $R/gpurun_out/r4/prof_${wl}_amp -name "*kernel_stats.csv" | head -1)
  echo "== $wl --amp"; tail -1 $R/gpurun_out/r4/prof_${wl}_amp.log | cut -c1-200
  python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{float(r['Percentage']):5.1f}%  {int(r['Calls']):6d} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:120]}")
P
  cp "$f" $R/gpurun_out/r4/${wl}_amp_kernel_stats.csv
done
