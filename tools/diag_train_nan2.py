"""Diagnostic (GPU): which FORWARD operand-range reduction, when done by osa_amax_f32 instead of torch, makes the captured GwcNet training
step replay NaN -- bisection over the call index inside one step."""
import argparse, os, sys, torch
os.environ["OSA_ENGINE_AMAX"] = "1"; os.environ["OSA_AMAX_MODE"] = "idx"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from openstereo_amd import engine, ranges
engine.set_precision("f16x3")
args = argparse.Namespace(batch=None, workload="gwcnet_train", steps=10, warmup=2, no_graph=False, streams=1, no_workloads=True, timed_only=True, gpus=1, precision="f16x3")

def trial(lo, hi, log=False):
    ranges.DIAG.update(lo=lo, hi=hi, log=[] if log else None)
    wl = bench.WORKLOADS["gwcnet_train"](args, torch.device("cuda:0"), 0)
    real_step = wl.step
    def step():
        ranges.DIAG["count"] = 0
        return real_step()
    wl.step = step
    for _ in range(2):
        wl.step()
    calls = list(ranges.DIAG["log"] or [])
    ranges.DIAG["log"] = None
    cap = bench.capture_training_step(wl)
    vals = [float(cap[1]()) for _ in range(4)] if cap else None
    del wl, cap
    torch.cuda.empty_cache()
    ok = vals is not None and all(v == v for v in vals)
    return ok, vals, calls

for lo, hi in ((0, 5), (0, 6), (0, 7), (0, 8), (1, 9), (2, 9), (3, 9), (9, 18), (18, 36), (4, 8), (5, 9)):
    ok, vals, _ = trial(lo, hi)
    print(f"kernel on forward calls [{lo}, {hi}):", "ok" if ok else "NaN", None if vals is None else [round(v, 3) for v in vals], flush=True)
