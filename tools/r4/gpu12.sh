#!/bin/bash
# round 4, GPU call 12: depthwise kernel v2 (chunked strips): parity + LightStereo kernel table
cd "$(dirname "$0")/../.."
R=$(pwd); mkdir -p gpurun_out/r4
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "depthwise or lightstereo" 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tail -3
for a in "" "--amp"; do
  echo "== lightstereo_kitti15 $a"; timeout 300 python bench.py --workload lightstereo_kitti15 $a --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof_ls2 -o p -- python $R/bench.py --workload lightstereo_kitti15 --no-graph --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r4/prof_ls2.log 2>&1
f=$(find $R/gpurun_out/r4/prof_ls2 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    print(f"{float(r['Percentage']):5.1f}%  {int(r['Calls']):6d} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}  {r['Name'][:100]}")
P
