"""Operator-level view of a training step (torch.profiler, one EAGER step after warm-up): which torch ops / engine launches the kernels of
tools/prof_train_graph.sh belong to, with input shapes.
    python tools/prof_train_ops.py [--workload stereobase_e2e_train] [--amp] [--top 60]"""
import argparse
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="stereobase_e2e_train")
ap.add_argument("--amp", action="store_true")
ap.add_argument("--top", type=int, default=60)
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--stacks", default="", help="comma-separated op names (aten::copy_,aten::add_,...): device time of each grouped by input shapes (and by the Python frames of this package where the profiler records them: ops issued by the autograd engine have none)")
a = ap.parse_args()
import bench  # noqa: E402
from openstereo_amd import engine  # noqa: E402

engine.set_precision("f16x3")
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS[a.workload](a, dev, 0)
wl.static = True
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=bool(a.stacks)) as prof:
    wl.step()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True)
rows = sorted(ev, key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"one eager {a.workload} step{' --amp' if a.amp else ''}: {tot / 1e3:.1f} ms of device time in {sum(e.count for e in rows if e.self_device_time_total > 0)} device-active op calls")
for e in rows[:a.top]:
    if e.self_device_time_total <= 0:
        break
    print(f"{100 * e.self_device_time_total / tot:5.1f} %  {e.self_device_time_total / 1e3:8.2f} ms  x{e.count:5d}  {e.key[:60]:60s} {str(e.input_shapes)[:150]}")
# by op name only
ev2 = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
print("\n-- by op name --")
for e in ev2[:40]:
    if e.self_device_time_total <= 0:
        break
    print(f"{100 * e.self_device_time_total / tot:5.1f} %  {e.self_device_time_total / 1e3:8.2f} ms  x{e.count:5d}  {e.key[:90]}")
if a.stacks:
    want = set(a.stacks.split(","))
    groups = {}
    for e in prof.events():
        if e.name in want and e.device_time_total > 0:
            st = [f for f in (e.stack or []) if "/openstereo_amd/" in f or "bench.py" in f][:4]
            key = (e.name, str(e.input_shapes)[:70], " <- ".join(s_.split("/")[-1] for s_ in st) or "(autograd engine / no Python frame)")
            g = groups.setdefault(key, [0, 0.0])
            g[0] += 1; g[1] += e.device_time_total
    print("\n-- by call stack --")
    for (name, shp, st), (n, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:90]:
        print(f"{t / 1e3:8.2f} ms  x{n:5d}  {name:16s} {shp:70s} {st}")
