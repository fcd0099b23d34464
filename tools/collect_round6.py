"""Copy the summaries of the round-6 final pass (gpurun_out/r6_final, gpurun_out/prof_r6: tools/profile_round6.sh; gpurun_out/r6: the
training-step kernel table of tools/r6/final_train_table.sh) into profiles/round6/ and merge the per-launch HBM traffic of that pass into
profiles/traffic.json, stamped with the commit it was measured at.
    python tools/collect_round6.py"""
import csv
import json
import os
import shutil
import subprocess

R = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles", "round6")


def cp(src, dst):
    src = os.path.join(G, src)
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("MISSING", src)


cp("r6_final/bench_default.json", "r6_bench_default_final.json")
cp("r6_final/bench_time.txt", "r6_final_bench_wall_time.txt")
cp("prof_r6/kernel_stats.csv", "r6_final_kernel_stats.csv")
cp("prof_r6/sq_summary.txt", "r6_final_sq_summary.txt")
cp("r6_final/determinism.txt", "r6_final_determinism.txt")
cp("r6/train_amp_kernels_final.txt", "stereobase_e2e_train_amp_replay_kernel_table_final.txt")
cp("r6/suite.txt", "r6_gpu_suite_tail.txt")

commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, cwd=R).stdout.strip()
bench = json.loads(open(os.path.join(G, "r6_final", "bench_default.json")).read().strip().splitlines()[-1])

# every launch of the dominant kernel in the one-stream trace
rows = sorted(csv.DictReader(open(os.path.join(G, "prof_r6", "trace", "bench_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda sub: [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if sub in r["Kernel_Name"]]
m, s2 = dur("conv_march_kernel<4, 16, 1, 1>"), dur("conv_march_s2_kernel<8>")
short = [x for x in m if x < 3.5]                       # the 32 -> 32 launches (the 64 -> 32 dres0.0 call of the same instance takes ~4.4 ms)
ms = sum(short) / len(short)
with open(os.path.join(P, "r6_final_dominant_kernel_launches.txt"), "w") as f:
    f.write(f"rocprofv3 --kernel-trace of `bench.py --timed-only --no-graph --streams 1 --steps 5 --warmup 2 --batch 9` (tools/profile_round4.sh r6), round-6 final pass, code state of commit {commit}\n")
    f.write(f"conv_march_kernel<4, 16, 1, 1>: durations of its {len(m)} launches in ms (per step: dres0.0 64->32 -- the long one --, then dres0.2, dres1.0, dres1.2 32->32)\n")
    f.write(" ".join(f"{x:.3f}" for x in m) + "\n")
    f.write(f"mean of the {len(short)} 32->32 launches: {ms:.4f} ms -> 779.8 GFLOP / {ms:.4f} ms = {779.8 / ms:.1f} TFLOP/s = {779.8 / ms / 833.3:.3f} of 833 "
            f"(bench.py's HIP-event figure: {bench['roofline']['avg_launch_ms']} ms = {bench['roofline']['frac']:.3f})\n")
    if s2:
        t2 = sum(s2) / len(s2)
        f.write(f"conv_march_s2_kernel<8>: {len(s2)} launches, mean {t2:.4f} ms (584.8 / 3 GFLOP per launch -> {584.8 / 3 / t2:.1f} TFLOP/s = {584.8 / 3 / t2 / 833.3:.3f} of 833)\n")
print(open(os.path.join(P, "r6_final_dominant_kernel_launches.txt")).read())

# HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, tools/parse_pmc3.py) of this pass -> profiles/traffic.json keys *_B9
t = json.load(open(os.path.join(G, "prof_r6", "traffic.json")))
cur = json.load(open(os.path.join(R, "profiles", "traffic.json")))
for k, v in t.items():
    if k.endswith("_B9"):
        cur[k] = v
if "_detail" in t:
    cur["_detail_B9"] = t["_detail"]
cur["_measured_at"] = f"round 6 final pass, code state of commit {commit} (tools/profile_round6.sh; keys *_B9)"
json.dump(cur, open(os.path.join(R, "profiles", "traffic.json"), "w"), indent=1)
print("traffic.json:", {k: v for k, v in cur.items() if k.endswith("_B9") and not isinstance(v, dict)})
w = bench["workloads"]
print("headline", bench["value"], bench["ms_per_step"], bench["roofline"]["frac"], "| train amp", w["stereobase_e2e_train_amp"]["ms_per_step"], "f16x3", w["stereobase_e2e_train"]["ms_per_step"])
