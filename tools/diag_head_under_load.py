"""Is the fused head (csrc/softargmin.hip upsample4_softargmin_kernel) deterministic while MFMA-heavy launches run on other streams?  (r5)
    python tools/diag_head_under_load.py [--load f16x3|f32|none] [--iters N] [--kernel head|copy|classifier]
Reference outputs are computed on an idle GPU; then the same launch is repeated on one stream while two other streams loop a 3x3x3 32->32
convolution at 3 pairs; every output is compared bit for bit with the reference."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--load", default="f16x3")
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--kernel", default="head")
ap.add_argument("--tag", default="")
a = ap.parse_args()
from openstereo_amd import _lib, engine, ops  # noqa: E402
from openstereo_amd.engine import PackedConv3d  # noqa: E402
_lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
B, D, H, W = 3, 48, 136, 240
cost = (torch.randn(B, D, H, W, generator=g) * 3.0).to(dev)
x32 = ops.to_cl((torch.randn(B, 32, D, H, W, generator=g)).to(dev))
conv = nn.Conv3d(32, 32, 3, padding=1, bias=False).to(dev)
small = nn.Conv3d(32, 1, 3, padding=1, bias=False).to(dev)
fl, fr = torch.randn(B, 320, H, W, generator=g).to(dev), torch.randn(B, 320, H, W, generator=g).to(dev)
cl_, cr_ = torch.randn(B, 12, H, W, generator=g).to(dev), torch.randn(B, 12, H, W, generator=g).to(dev)


def launch():
    if a.kernel == "head":
        return ops.upsample_softargmin(cost, 192, 4 * H, 4 * W)
    if a.kernel == "copy":
        return cost.clone()
    if a.kernel == "classifier":
        return clf(x32)
    if a.kernel == "softmax":                       # softmax_softargmin_kernel (StereoBase / IGEV / LightStereo heads): 18 packed-fp32 instructions as shipped in r5
        return ops.softmax_disparity_regression(cost, D)
    if a.kernel == "upgeneric":                     # the any-size upsample + soft-argmin kernel (71 packed-fp32 instructions as shipped in r5)
        return ops.upsample_softargmin(cost, 96, 2 * H, 2 * W)
    if a.kernel == "gwcvol":                        # NCDHW volume builder
        return ops.build_gwc_volume(fl, fr, 48, 40)
    if a.kernel == "clvol":                         # the fused gwc + concat builder writing NDHWC (osa_build_volume_f32, layout 1)
        return ops.build_cost_volume_cl(fl, fr, 40, cl_, cr_, maxdisp=48)
    raise SystemExit("unknown kernel")


with torch.no_grad():
    clf = engine.SmallCoConv3d(small)
    loads = []
    if a.load != "none":
        split = a.load == "f16x3_split"          # the dominant instance of the model: split input AND split output (no scratch, 250 VGPRs)
        prec = "f16x3" if split else a.load
        pc0 = PackedConv3d(conv, None, 1, precision=prec)
        pc = (lambda t: pc0(t, out_split=True)) if split else pc0
        xin = [x32.clone() for _ in range(2)]
        for i, t in enumerate(xin):
            if prec == "f16x3":
                t._osa_meta = engine.input_meta(t)
            if split:
                xin[i] = pc0(t, out_split=True)
        torch.cuda.synchronize()
        loads = [(torch.cuda.Stream(), t) for t in xin]
        for st, t in loads:
            with torch.cuda.stream(st):
                pc(t)
    ref = launch().clone()
    torch.cuda.synchronize()
    idle_bad = 0
    for _ in range(10):
        idle_bad += int((launch().view(torch.int32) != ref.view(torch.int32)).sum())
    torch.cuda.synchronize()
    bad_total, bad_iters, outs = 0, 0, []
    for it in range(a.iters):
        for st, t in loads:
            with torch.cuda.stream(st):
                for _ in range(3):
                    pc(t)
        outs.append(launch())
    torch.cuda.synchronize()
    for it, o in enumerate(outs):
        neq = o.view(torch.int32) != ref.view(torch.int32)
        n = int(neq.sum())
        if n:
            bad_total += n
            bad_iters += 1
            if bad_iters <= 4:
                idx = neq.nonzero()
                last = idx[-1].tolist()
                print(f"  iter {it}: {n} differing elements, first {idx[0].tolist()} last {last}, max |diff| {float((o - ref).abs().max()):.4f}")
print(f"[{a.tag or a.kernel}] load={a.load}: idle mismatches {idle_bad}; under load {bad_total} differing elements in {bad_iters} of {a.iters} launches")
