"""dtypes / layouts of the tensors that travel through one GRU iteration of the StereoBase AMP training forward (where the dtype-copy kernels of
the step come from):  python tools/probe_amp_dtypes.py"""
import os, sys, argparse
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench
from openstereo_amd import engine
from openstereo_amd.models import igev_update as IU
engine.set_precision("f16x3")
a = argparse.Namespace(batch=None, amp=True, steps=1, warmup=0, precision="f16x3", workload="stereobase_e2e_train", streams=None)
wl = bench.WORKLOADS["stereobase_e2e_train"](a, torch.device("cuda", 0), 0)
seen = set()
def desc(t):
    if not isinstance(t, torch.Tensor):
        return str(type(t).__name__)
    cl = "cl" if (t.dim() == 4 and t.stride(1) == 1 and t.shape[1] > 1) else ("nchw" if t.is_contiguous() else "strided")
    return f"{str(t.dtype).replace('torch.', '')}[{','.join(map(str, t.shape))}]{cl}"
orig = IU.ConvGRU.forward_train
def ft(self, h, cz, cr, cq, *x_list):
    out = orig(self, h, cz, cr, cq, *x_list)
    key = (h.shape[-1], len(seen) // 3)
    if len(seen) < 6:
        seen.add(key)
        print("ConvGRU", "h", desc(h), "cz", desc(cz), "cq", desc(cq), "x", [desc(x) for x in x_list], "->", desc(out))
    return out
IU.ConvGRU.forward_train = ft
oe = IU.BasicMotionEncoder.forward_train
cnt = [0]
def fe(self, disp, corr):
    o = oe(self, disp, corr)
    if cnt[0] < 2:
        cnt[0] += 1
        print("encoder disp", desc(disp), "corr", desc(corr), "->", desc(o))
    return o
IU.BasicMotionEncoder.forward_train = fe
ou = IU.BasicMultiUpdateBlock.forward_train
c2 = [0]
def fu(self, net, inp, corr=None, disp=None, iter04=True, iter08=True, iter16=True, update=True):
    r = ou(self, net, inp, corr, disp, iter04, iter08, iter16, update)
    if c2[0] < 2 and update:
        c2[0] += 1
        print("update: net", [desc(t) for t in net], "-> net", [desc(t) for t in r[0]], "mask", desc(r[1]), "delta", desc(r[2]))
    return r
IU.BasicMultiUpdateBlock.forward_train = fu
wl.step()
torch.cuda.synchronize()
