"""Diagnostic (GPU): bitwise repeatability of the engine's training convolutions (forward, dgrad, wgrad) on the update block's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openstereo_amd import autograd as AG, engine
torch.manual_seed(0)
for prec in ("f32", "f16x3"):
    for (Ci, Co, H, W, k) in ((256, 128, 4, 8, 3), (384, 128, 8, 16, 3), (384, 128, 16, 32, 3), (128, 256, 16, 32, 3), (36, 64, 16, 32, 1), (64, 64, 16, 32, 3),
                              (128, 127, 16, 32, 3), (64, 64, 64, 128, 3), (96, 96, 32, 64, 3), (128, 128, 16, 32, 3), (128, 128, 4, 8, 3)):
        for fmt in ("nchw", "cl"):
            x = torch.randn(1, Ci, H, W, device="cuda")
            if fmt == "cl":
                x = x.contiguous(memory_format=torch.channels_last)
            w = (torch.randn(Co, Ci, k, k, device="cuda") * 0.05).requires_grad_(True)
            b = torch.randn(Co, device="cuda").requires_grad_(True)
            dy = torch.randn(1, Co, H, W, device="cuda")
            outs = []
            for rep in range(40):
                xx = x.clone().requires_grad_(True)
                junk = torch.full((1 << 20,), float(rep), device="cuda"); del junk
                y = AG.conv2d(xx, w, b, 1, k // 2, 1, precision=prec)
                gx, gw = torch.autograd.grad(y, (xx, w), dy)
                outs.append((y.detach().clone(), gx.clone(), gw.clone()))
            nd = [sum(not torch.equal(o[i], outs[0][i]) for o in outs) for i in range(3)]
            md = [max(float((o[i] - outs[0][i]).abs().max()) for o in outs) for i in range(3)]
            flag = "" if not any(nd) else "   <<< NOT REPEATABLE"
            print(f"{prec:6s} {Ci:4d}->{Co:<4d} {H}x{W} k{k} x {fmt:4s}: runs differing from run 0 (of 39): y {nd[0]} dx {nd[1]} dw {nd[2]}; max abs diff {md}{flag}")
