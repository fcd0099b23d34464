"""In-kernel timeline of the fused transposed conv (or any conv launch): experiments build + OSA_DBG=256.

    OSA_LIB_PATH=.../exp.so python tools/trace_conv.py [--batch 2] [--layer conv6|conv5|c32]

Prints, per traced workgroup (every 97th) and wave, the microseconds between the phase stamps of conv_kernel.h:
0 kernel entry | per staging pass: pass start, own loads landed, brick complete (barrier), taps done | 20 epilogue start (barrier) | 21+i tile i done."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from openstereo_amd import _lib, engine  # noqa: E402
from openstereo_amd.engine import PackedConv3d, ACT_NONE, ACT_RELU  # noqa: E402
from bench_deconv import split_of  # noqa: E402

SLOTS, WAVES, EVENTS = 32, 4, 40


def read_trace():
    lib = _lib.load()
    buf = np.zeros(SLOTS * WAVES * EVENTS, dtype=np.uint64)
    fn = lib.osa_debug_trace_read
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    fn.restype = ctypes.c_int
    assert fn(buf.ctypes.data, buf.size) == buf.size
    return buf.reshape(SLOTS, WAVES, EVENTS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--layer", default="conv6")
    ap.add_argument("--slots", type=int, default=6)
    args = ap.parse_args()
    dev = "cuda:0"
    engine.set_precision("f16x3")
    V0, V1, V2 = (48, 136, 240), (24, 68, 120), (12, 34, 60)
    if args.layer in ("l1", "l2", "l3"):            # GwcNet backbone layers (D = 1), 2 images per pair, residual block tail
        C, dims = {"l1": (32, (1, 272, 480)), "l2": (64, (1, 136, 240)), "l3": (128, (1, 136, 240))}[args.layer]
        xs, _ = split_of(C, dims, 2 * args.batch, dev)
        rs, _ = split_of(C, dims, 2 * args.batch, dev)
        cv = PackedConv3d(nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev), nn.BatchNorm2d(C).to(dev).eval(), ACT_NONE)
        fn = lambda: cv(xs, residual=rs, out_split=True)
    elif args.layer == "c32":
        xs, _ = split_of(32, V0, args.batch, dev)
        cv = PackedConv3d(nn.Conv3d(32, 32, 3, 1, 1, bias=False).to(dev), nn.BatchNorm3d(32).to(dev).eval(), ACT_RELU)
        fn = lambda: cv(xs, out_split=True)
    else:
        Ci, Co, din, dout = (64, 32, V1, V0) if args.layer == "conv6" else (128, 64, V2, V1)
        xs, _ = split_of(Ci, din, args.batch, dev)
        rs, _ = split_of(Co, dout, args.batch, dev)
        dc = PackedConv3d(nn.ConvTranspose3d(Ci, Co, 3, stride=2, padding=1, output_padding=1, bias=False).to(dev), nn.BatchNorm3d(Co).to(dev).eval(), ACT_RELU)
        rl = PackedConv3d(nn.Conv3d(Co, Co, 1, bias=False).to(dev), nn.BatchNorm3d(Co).to(dev).eval(), ACT_NONE)
        fn = lambda: dc(xs, redir=(rl, rs), out_split=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    read_trace()                                   # clear
    os.environ["OSA_DBG"] = "256"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    os.environ.pop("OSA_DBG")
    t = read_trace().astype(np.int64)
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us")
    t0 = t[t > 0].min()
    for s in range(args.slots):
        for w in range(WAVES):
            ev = [(i, (t[s, w, i] - t0) / 100.0) for i in range(EVENTS) if t[s, w, i]]
            if not ev:
                continue
            line = f"wg {s * 97:5d} wave {w}: start {ev[0][1]:7.2f} |"
            for (i0, a), (i1, b) in zip(ev, ev[1:]):
                line += f" {i1}:+{b - a:.2f}"
            print(line + f" | total {ev[-1][1] - ev[0][1]:.2f} us")


if __name__ == "__main__":
    main()
