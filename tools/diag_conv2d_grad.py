"""Diagnostic (GPU): engine conv2d / conv_transpose2d autograd (forward, dgrad, wgrad) vs stock PyTorch-ROCm over the update block's shapes."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openstereo_amd import autograd as AG, engine        # noqa: E402

torch.manual_seed(0)
dev = "cuda"
cases = []
for (H, W) in ((4, 8), (8, 16), (16, 32), (5, 9), (20, 46), (34, 60)):
    for (Ci, Co, k) in ((256, 128, 3), (384, 128, 3), (128, 256, 3), (128, 32, 3), (256, 1, 3), (164, 64, 1), (64, 64, 3), (128, 127, 3), (96, 96, 1), (96, 96, 3)):
        cases.append((H, W, Ci, Co, k))
for prec in ("f32", "f16x3"):
    engine.set_precision(prec)
    print("precision", prec)
    for (H, W, Ci, Co, k) in cases:
        x = torch.randn(1, Ci, H, W, device=dev)
        w = torch.randn(Co, Ci, k, k, device=dev) * (1.0 / (Ci * k * k) ** 0.5)
        b = torch.randn(Co, device=dev) * 0.1
        gy = torch.randn(1, Co, H, W, device=dev)
        res = []
        for fn in (lambda xx, ww, bb: F.conv2d(xx, ww, bb, 1, k // 2), lambda xx, ww, bb: AG.conv2d(xx, ww, bb, 1, k // 2)):
            xx, ww, bb = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
            y = fn(xx, ww, bb)
            (y * gy).sum().backward()
            res.append((y.detach(), xx.grad, ww.grad, bb.grad))
        e = [float((a - r).abs().max() / (r.abs().max() + 1e-20)) for r, a in zip(*res)]
        flag = "  <-----" if max(e) > (1e-4 if prec == "f32" else 1e-3) else ""
        print(f"  {H:3d}x{W:<3d} {Ci:4d}->{Co:<4d} k{k}:  y {e[0]:.1e}  dx {e[1]:.1e}  dw {e[2]:.1e}  db {e[3]:.1e}{flag}")
