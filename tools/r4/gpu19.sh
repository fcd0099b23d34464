#!/bin/bash
# round 4, GPU call 19: final profile passes (kernel stats, PMC traffic, SQ) at the new default batch + the default bench line with workloads
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
echo "== profile"
bash tools/profile_round4.sh r4b > gpurun_out/r4/profile_b.log 2>&1; tail -3 gpurun_out/r4/profile_b.log; head -c 1500 gpurun_out/prof_r4b/traffic.json; echo; cat gpurun_out/prof_r4b/sq_summary.txt 2>/dev/null | head -40
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
echo "== default bench (shipped lib) with workloads"
timeout 1500 python bench.py 2>gpurun_out/r4/bench19.err | tail -1 > gpurun_out/r4/bench19.json; python - <<'P'
import json
d=json.load(open('gpurun_out/r4/bench19.json'))
print({k:d.get(k) for k in ('value','ms_per_step','eager_value','eager_ms_per_step')}); print(d['config'].get('pairs_per_gpu_per_step'), d['config'].get('sub_batch_streams'), d['config'].get('latency_ms_1_pair'))
print(d['roofline'])
for r in d['rooflines']: print({k:r.get(k) for k in ('kernel','achieved','frac','avg_launch_ms','traffic')})
print(d['cpu_baseline'].get('value'), d['cpu_baseline'].get('pytorch_rocm_eager_same_gpu',{}).get('value'), d.get('other_precision',{}).get('value'))
for k,v in d.get('workloads',{}).items(): print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','eager_value','error','skipped')}, (v.get('pytorch_rocm_eager_same_gpu') or {}).get('value'))
P
tail -5 gpurun_out/r4/bench19.err
