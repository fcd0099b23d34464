"""Diagnostic (GPU): losses of the first steps of a training workload, eager and as a replayed hipGraph.
    [OSA_ENGINE_AMAX=1 [OSA_AMAX_MODE=fwd|bwd|both]] python tools/diag_train_nan.py [gwcnet_train]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from openstereo_amd import engine
name = sys.argv[1] if len(sys.argv) > 1 else "gwcnet_train"
engine.set_precision("f16x3")
args = argparse.Namespace(batch=None, workload=name, steps=10, warmup=2, no_graph=False, streams=1, no_workloads=True, timed_only=True, gpus=1, precision="f16x3")
wl = bench.WORKLOADS[name](args, torch.device("cuda:0"), 0)
print("eager:", [round(float(wl.step()), 4) for _ in range(8)])
cap = bench.capture_training_step(wl)
if cap:
    print("graph:", [round(float(cap[1]()), 4) for _ in range(16)])
