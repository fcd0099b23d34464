"""Calibration launches for the FETCH_SIZE / WRITE_SIZE counters (run under `rocprofv3 --pmc ...`, tools/profile_round3.sh): the engine's
depthwise kernel with a 1x1 identity filter over an 8.6 GB NHWC tensor (far beyond the 256 MiB Infinity Cache), once reading ALL 32
channels of every pixel (full 128-byte lines, 16 B per lane) and once reading 16 of the 32 (64-byte segments at a 128-byte stride -- the
access pattern of the conv engine's brick staging and of the classifier).  Algorithmic bytes are known exactly, so the ratio
counter / bytes calibrates the x2 correction of MI355X_MICROARCH.md for both patterns; tools/parse_pmc3.py stores it as `_calibration`."""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openstereo_amd import ops                      # noqa: E402
from openstereo_amd.engine import DepthwiseConv2d   # noqa: E402

H = W = 8192
x = ops.empty_cl(1, 32, 1, H, W, "cuda")
x.normal_()
for C in (32, 16):
    conv = nn.Conv2d(C, C, 1, groups=C, bias=False).cuda()
    with torch.no_grad():
        conv.weight.fill_(1.0)
    dw = DepthwiseConv2d(conv)
    for _ in range(3):
        y = dw(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        y = dw(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    rd, wr = H * W * C * 4, H * W * C * 4
    print(f"calibration C={C} of 32: reads {rd / 1e9:.3f} GB (useful), writes {wr / 1e9:.3f} GB, {ms:.3f} ms = {(rd + wr) / ms / 1e6:.0f} GB/s useful", flush=True)
    del y
