"""A/B of the stage->barrier->taps conv kernel against its persistent LDS-DMA pipelined form on the dominant layer
(3x3x3 32->32 @ 48x136x240, split input / split output), with timing-only ablations.  Needs the experiments build:

    tools/build_variant.sh exp -DOSA_EXPERIMENTS
    OSA_LIB_PATH=openstereo_amd/lib/variants/exp.so python tools/bench_pipe.py [--batch 2] [--iters 20]

Every variant runs interleaved in ONE process (rounds x variants), medians are reported (cdna_hip_programming.md 5.4 rule 24)."""
import argparse
import os
import statistics
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from openstereo_amd import ops  # noqa: E402
from openstereo_amd.engine import PackedConv3d  # noqa: E402

VARIANTS = [  # name, env
    ("classic", {}),
    ("classic no-ranges", {"BENCH_NO_RANGES": "1"}),
    ("pipe", {"OSA_PIPE": "1"}),
    ("pipe 1wg/cu", {"OSA_PIPE": "1", "OSA_PIPE_WGS": "1"}),
    ("pipe no-dma", {"OSA_PIPE": "1", "OSA_DBG": "1"}),
    ("pipe no-epilogue", {"OSA_PIPE": "1", "OSA_DBG": "8"}),
    ("pipe no-taps", {"OSA_PIPE": "1", "OSA_DBG": "16"}),
    ("pipe taps-only", {"OSA_PIPE": "1", "OSA_DBG": "9"}),
    ("pipe dma-only", {"OSA_PIPE": "1", "OSA_DBG": "24"}),
    ("classic no-staging", {"OSA_DBG": "1"}),
    ("classic no-epilogue", {"OSA_DBG": "8"}),
]
KEYS = ("OSA_PIPE", "OSA_PIPE_WGS", "OSA_DBG", "BENCH_NO_RANGES")


def no_ranges(layer, xs):
    """The same launch with a NULL osa_f16x3_ranges block: no operand scales, no max |y| tracking (cost of the range machinery)."""
    from openstereo_amd import _lib
    orig = _lib.F16x3Ranges
    _lib.F16x3Ranges = lambda *a: None
    try:
        return layer(xs, out_split=True)
    finally:
        _lib.F16x3Ranges = orig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--ci", type=int, default=32)
    ap.add_argument("--dims", default="48,136,240")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = "cuda:0"
    D, H, W = (int(v) for v in args.dims.split(","))
    pre = nn.Conv3d(args.ci, args.ci, 1, bias=False).to(dev)
    conv = nn.Conv3d(args.ci, 32, 3, 1, 1, bias=False).to(dev)
    x = ops.empty_cl(args.batch, args.ci, D, H, W, dev)
    x.normal_()
    xs = PackedConv3d(pre, None, 1, precision="f16x3")(x, out_split=True)
    layer = PackedConv3d(conv, nn.BatchNorm3d(32).to(dev).eval(), 1, precision="f16x3")
    gflop = 2 * 27 * args.ci * 32 * D * H * W * args.batch / 1e9
    variants = [v for v in VARIANTS if not args.only or args.only in v[0]]
    times = {n: [] for n, _ in variants}
    for r in range(args.rounds + 1):
        for name, env in variants:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            run = (lambda: layer(xs, out_split=True)) if "BENCH_NO_RANGES" not in env else (lambda: no_ranges(layer, xs))
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            if r:                       # round 0 = warm-up
                times[name].append(e0.elapsed_time(e1) / args.iters)
    for k in KEYS:
        os.environ.pop(k, None)
    print(f"3x3x3 {args.ci}->32 @{D}x{H}x{W}, {args.batch} pairs per launch: {gflop:.1f} algorithmic GFLOP")
    for name, _ in variants:
        t = times[name]
        med = statistics.median(t)
        print(f"  {name:22s} median {med:7.4f} ms  min {min(t):7.4f}  {gflop / med:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
