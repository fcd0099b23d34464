"""Copy the summaries of the round-5 final pass (gpurun_out/r5_final, gpurun_out/prof_r5: tools/profile_round5.sh) into profiles/round5/ and
merge the per-launch HBM traffic of that pass into profiles/traffic.json, stamped with the commit it was measured at.
    python tools/collect_round5.py"""
import json
import os
import re
import shutil
import subprocess

R = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles", "round5")
os.makedirs(P, exist_ok=True)


def cp(src, dst):
    src = os.path.join(G, src)
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("MISSING", src)


cp("prof_r5/kernel_stats.csv", "r5_final_kernel_stats_B9_1stream.csv")
cp("prof_r5/sq_summary.txt", "r5_final_pmc_sq_summary_B9.txt")
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
    cp(f"prof_r5/pmc_{c}/p_counter_collection.csv", f"r5_final_pmc_{c}_B9.csv")
cp("r5_final/determinism.txt", "final_determinism_checks.txt")
cp("r5_final/census.txt", "final_ctypes_census.txt")
# suite log: the summary lines only (library chatter removed)
s = os.path.join(G, "r5_final", "suite.log")
if os.path.exists(s):
    keep = [l for l in open(s, errors="replace") if not l.startswith(("GridwiseOp", "MIOpen(HIP)"))]
    open(os.path.join(P, "final_gpu_suite.txt"), "w").writelines(keep[-40:])
    print("copied final_gpu_suite.txt:", keep[-1].strip())
# the default bench line: exactly one JSON object (stdout of `python bench.py --gpus 1 --steps 20 --warmup 5`)
b = os.path.join(G, "r5_final", "bench_default.json")
if os.path.exists(b):
    lines = [l for l in open(b, errors="replace").read().splitlines() if l.strip()]
    js = [l for l in lines if l.startswith("{")]
    print(f"bench stdout: {len(lines)} non-empty line(s), {len(js)} JSON line(s)")
    d = json.loads(js[-1])
    json.dump(d, open(os.path.join(P, "r5_bench_default_final.json"), "w"), indent=1)
    print("headline", d["value"], d["unit"], d["ms_per_step"], "ms/step; roofline frac", d["roofline"]["frac"], "cpu_baseline", d.get("cpu_baseline", {}).get("value"))
    for k, v in (d.get("workloads") or {}).items():
        print(f"   {k}: {v.get('value')} {v.get('unit')} ({v.get('ms_per_step')} ms)")
cp("r5_final/bench_time.txt", "final_bench_wall_time.txt")
cp("r5_final/power_summary.txt", "final_power_and_clock_timed_config.txt")
# traffic
t = os.path.join(G, "prof_r5", "traffic.json")
if os.path.exists(t):
    new, cur = json.load(open(t)), json.load(open(os.path.join(R, "profiles", "traffic.json")))
    head = subprocess.check_output(["git", "-C", R, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    for k, v in new.items():
        if not k.startswith("_"):
            cur[k] = v
    cur["_detail_B9"] = new.get("_detail", cur.get("_detail_B9"))
    cur["_measured_at"] = f"round 5 final pass, code state of commit {head} (tools/profile_round5.sh; keys *_B9)"
    json.dump(cur, open(os.path.join(R, "profiles", "traffic.json"), "w"), indent=1)
    print("traffic.json:", {k: v for k, v in new.items() if not k.startswith("_")})
