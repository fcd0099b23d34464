"""Timing-only ablations of the fused stride-2 transposed conv (GwcNet hourglass conv6 + redir1 / conv5 + redir2) in the form the
model runs it: split input, split redir input, split output.  GPU only; needs the experiments build for OSA_DBG
(tools/build_variant.sh exp -DOSA_EXPERIMENTS; OSA_LIB_PATH=.../exp.so).

    python tools/bench_deconv.py [--batch 2] [--dbgs 0,8,32,64,1]
"""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from openstereo_amd import ops, ranges, engine  # noqa: E402
from openstereo_amd.engine import PackedConv3d, ACT_NONE, ACT_RELU  # noqa: E402


def split_of(C, dims, B, dev):
    """A split tensor with C channels: output of a 1x1x1 engine conv."""
    x = ops.empty_cl(B, C, *dims, dev)
    x.normal_()
    ranges.ensure_meta(x)
    ident = PackedConv3d(nn.Conv3d(C, C, 1, bias=False).to(dev), None, ACT_NONE)
    return ident(x, out_split=True), x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dbgs", default="0,8,32,64,1,9")
    args = ap.parse_args()
    dev = "cuda:0"
    engine.set_precision("f16x3")
    V0, V1, V2 = (48, 136, 240), (24, 68, 120), (12, 34, 60)
    for name, Ci, Co, din, dout in (("conv6+redir1 64->32 V1->V0", 64, 32, V1, V0), ("conv5+redir2 128->64 V2->V1", 128, 64, V2, V1)):
        xs, xp = split_of(Ci, din, args.batch, dev)
        rs, rp = split_of(Co, dout, args.batch, dev)
        dc = PackedConv3d(nn.ConvTranspose3d(Ci, Co, 3, stride=2, padding=1, output_padding=1, bias=False).to(dev), nn.BatchNorm3d(Co).to(dev).eval(), ACT_RELU)
        rl = PackedConv3d(nn.Conv3d(Co, Co, 1, bias=False).to(dev), nn.BatchNorm3d(Co).to(dev).eval(), ACT_NONE)
        out_bytes = args.batch * Co * dout[0] * dout[1] * dout[2] * 4
        in_bytes = args.batch * Ci * din[0] * din[1] * din[2] * 4
        forms = {"fused split": lambda: dc(xs, redir=(rl, rs), out_split=True),
                 "plain (no redir, fp32 tensors)": lambda: dc(xp)}
        for fname, fn in forms.items():
            traffic = in_bytes + out_bytes * (2 if "fused" in fname else 1)
            line = f"{name:28s} {fname:32s}"
            for d in [int(v) for v in args.dbgs.split(",")]:
                os.environ["OSA_DBG"] = str(d)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.iters
                line += f" | dbg {d}: {ms:6.3f} ms"
                if d == 0:
                    line += f" ({traffic / ms / 1e9:5.2f} TB/s)"
            os.environ.pop("OSA_DBG", None)
            print(line, flush=True)


if __name__ == "__main__":
    main()
