"""Weight-gradient kernel (csrc/wgrad.hip) on the layer shapes of the training workloads; experiments build: sweeps OSA_WGRAD_TD
(position brick 2x8x8 / 4x8x8) and OSA_WGRAD_STRIP (w-bricks per workgroup; 0 = the host's choice).  GPU only.

    OSA_LIB_PATH=.../exp.so python tools/bench_wgrad.py [--strips 0,16,8,4,2] [--tds 2,4]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from openstereo_amd import _lib, ops  # noqa: E402

# name, transposed, Ci, Co, k, stride, input dims
G0, G1, G2 = (48, 64, 128), (24, 32, 64), (12, 16, 32)            # GwcNet training crop 256x512: D/4 x H/4 x W/4
S0, S1, S2, S3 = (48, 80, 184), (24, 40, 92), (12, 20, 46), (6, 10, 23)   # StereoBase crop 320x736
SHAPES = [
    ("gwc 32->32 s1 V0", 0, 32, 32, 3, 1, G0),
    ("gwc 64->32 s1 V0", 0, 64, 32, 3, 1, G0),
    ("gwc 32->64 s2 V0", 0, 32, 64, 3, 2, G0),
    ("gwc 64->64 s1 V1", 0, 64, 64, 3, 1, G1),
    ("gwc 64->128 s2 V1", 0, 64, 128, 3, 2, G1),
    ("gwc 128->128 s1 V2", 0, 128, 128, 3, 1, G2),
    ("gwc deconv 128->64 k3 V2", 1, 128, 64, 3, 2, G2),
    ("gwc deconv 64->32 k3 V1", 1, 64, 32, 3, 2, G1),
    ("sb 24->48 s2 V0", 0, 24, 48, 3, 2, S0),
    ("sb 48->48 s1 V1", 0, 48, 48, 3, 1, S1),
    ("sb 96->96 s1 V2", 0, 96, 96, 3, 1, S2),
    ("sb deconv 48->24 k4 V1", 1, 48, 24, 4, 2, S1),
    ("sb deconv 96->48 k4 V2", 1, 96, 48, 4, 2, S2),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--strips", default="0")
    ap.add_argument("--tds", default="0")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--atomics", dest="two_stage", action="store_false", help="the one-stage float-atomics form")
    args = ap.parse_args()
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    for name, tr, Ci, Co, k, s, dims in SHAPES:
        if args.only and args.only not in name:
            continue
        D, H, W = dims
        if tr:
            Do, Ho, Wo = (2 * D, 2 * H, 2 * W)
            pad = 1
        else:
            Do, Ho, Wo = [(n + 2 - 3) // s + 1 for n in dims]
            pad = 1
        x = ops.empty_cl(1, Ci, D, H, W, dev); x.normal_()
        dy = ops.empty_cl(1, Co, Do, Ho, Wo, dev); dy.normal_()
        dw = torch.empty((Ci, Co, k, k, k) if tr else (Co, Ci, k, k, k), device=dev)
        pos = (D * H * W) if tr else (Do * Ho * Wo)
        gflop = 2.0 * pos * Ci * Co * k ** 3 / 1e9
        line = f"{name:28s} {gflop:7.2f} GFLOP"
        for td in [int(v) for v in args.tds.split(",")]:
            for strip in [int(v) for v in args.strips.split(",")]:
                os.environ.pop("OSA_WGRAD_TD", None); os.environ.pop("OSA_WGRAD_STRIP", None)
                if td:
                    os.environ["OSA_WGRAD_TD"] = str(td)
                if strip:
                    os.environ["OSA_WGRAD_STRIP"] = str(strip)

                dims = (1, D, H, W, Ci, Do, Ho, Wo, Co, k, k, k, s, pad, pad, pad, 1, 1, 1, tr)
                need = _lib.load().osa_conv3d_wgrad_workspace_bytes(*dims) if args.two_stage else 0
                ws = torch.empty(max(need // 4, 1), device=dev)

                def run():
                    if args.two_stage:
                        _lib.call("osa_conv3d_wgrad_ws_f32", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1, D, H, W, Ci, x.shape[1],
                                  Do, Ho, Wo, Co, dy.shape[1], k, k, k, s, pad, pad, pad, 1, 1, 1, tr, ws.data_ptr(), need, st)
                    else:
                        _lib.call("osa_conv3d_wgrad_f32", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1, D, H, W, Ci, x.shape[1],
                                  Do, Ho, Wo, Co, dy.shape[1], k, k, k, s, pad, pad, pad, 1, 1, 1, tr, st)
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    run()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.iters
                line += f" | td{td} s{strip}: {ms:6.3f} ms {gflop / ms:5.1f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
