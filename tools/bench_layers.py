"""Per-layer timing of the GwcNet 3-D aggregation shapes on the engine, optionally sweeping the
kernel tile configuration (OSA_CONV_CFG).  GPU only.

    python tools/bench_layers.py [--cfgs 0,1,3,10] [--iters 20] [--batch 1]
"""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from openstereo_amd import ops, ranges  # noqa: E402
from openstereo_amd.engine import PackedConv3d, SmallCoConv3d  # noqa: E402

V0, V1, V2 = (48, 136, 240), (24, 68, 120), (12, 34, 60)
LAYERS = [  # name, kind, Ci, Co, k, stride, input dims, count per forward
    ("dres0.0 64->32 V0", "conv", 64, 32, 3, 1, V0, 1),
    ("32->32 V0", "conv", 32, 32, 3, 1, V0, 4),
    ("conv1 32->64 s2", "conv", 32, 64, 3, 2, V0, 3),
    ("conv2 64->64 V1", "conv", 64, 64, 3, 1, V1, 3),
    ("conv3 64->128 s2", "conv", 64, 128, 3, 2, V1, 3),
    ("conv4 128->128 V2", "conv", 128, 128, 3, 1, V2, 3),
    ("conv5 deconv 128->64", "deconv", 128, 64, 3, 2, V2, 3),
    ("conv6 deconv 64->32", "deconv", 64, 32, 3, 2, V1, 3),
    ("redir1 1x1 32->32 V0", "conv", 32, 32, 1, 1, V0, 3),
    ("redir2 1x1 64->64 V1", "conv", 64, 64, 1, 1, V1, 3),
    ("classif 32->1 V0", "small", 32, 1, 3, 1, V0, 1),
]


# GwcNet feature extractor (2 images per pair, D = 1): name, kind, Ci, Co, k, stride, dims, count, dilation
F2, F4 = (1, 272, 480), (1, 136, 240)
LAYERS_2D = [
    ("first 32->32 half", "conv2d", 32, 32, 3, 1, F2, 2 + 6, 1),
    ("l2.0 32->64 s2", "conv2d", 32, 64, 3, 2, F2, 1, 1),
    ("l2 64->64 quarter", "conv2d", 64, 64, 3, 1, F4, 31, 1),
    ("l3.0 64->128 quarter", "conv2d", 64, 128, 3, 1, F4, 1, 1),
    ("l3 128->128 quarter", "conv2d", 128, 128, 3, 1, F4, 5, 1),
    ("l4 128->128 dil2", "conv2d", 128, 128, 3, 1, F4, 6, 2),
    ("last 320->128", "conv2d", 320, 128, 3, 1, F4, 1, 1),
    ("last 128->12 1x1", "conv2d", 128, 12, 1, 1, F4, 1, 1),
]


# IGEV / StereoBase update block at 544x960 (one image per pair): 3x3 GRU gate convs on the 1/4, 1/8, 1/16 maps + the heads
G4, G8, G16 = (1, 136, 240), (1, 68, 120), (1, 34, 60)
LAYERS_GRU = [
    ("gru16 256->128 @34x60", "conv2d", 256, 128, 3, 1, G16, 9 * 32, 1),
    ("gru08 384->128 @68x120", "conv2d", 384, 128, 3, 1, G8, 6 * 32, 1),
    ("gru04 384->128 @136x240", "conv2d", 384, 128, 3, 1, G4, 3 * 32, 1),
    ("gru16 r|z 256->256 @34x60", "conv2d", 256, 256, 3, 1, G16, 3 * 32, 1),
    ("gru08 r|z 384->256 @68x120", "conv2d", 384, 256, 3, 1, G8, 2 * 32, 1),
    ("gru04 r|z 384->256 @136x240", "conv2d", 384, 256, 3, 1, G4, 32, 1),
    ("enc convc1 164->64 1x1", "conv2d", 164, 64, 1, 1, G4, 32, 1),
    ("enc conv 128->127", "conv2d", 128, 127, 3, 1, G4, 32, 1),
    ("head 128->256", "conv2d", 128, 256, 3, 1, G4, 32, 1),
    ("mask 128->32", "conv2d", 128, 32, 3, 1, G4, 32, 1),
    # LightStereo-S aggregation at 384x1248 (KITTI15): 1x1 expand / project convs of the MobileV2 blocks
    ("ls 48->192 @96x312", "conv2d", 48, 192, 1, 1, (1, 96, 312), 3, 1),
    ("ls 192->48 @96x312", "conv2d", 192, 48, 1, 1, (1, 96, 312), 3, 1),
    ("ls 96->384 @48x156", "conv2d", 96, 384, 1, 1, (1, 48, 156), 3, 1),
    ("ls 384->96 @48x156", "conv2d", 384, 96, 1, 1, (1, 48, 156), 3, 1),
    ("ls 192->768 @24x78", "conv2d", 192, 768, 1, 1, (1, 24, 78), 3, 1),
    ("ls 768->192 @24x78", "conv2d", 768, 192, 1, 1, (1, 24, 78), 3, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--only", default="")
    ap.add_argument("--set", default="3d", choices=["3d", "2d", "gru"], help="2d: the feature extractor's layers (batch = 2 x --batch images); gru: update-block layers")
    ap.add_argument("--env", default="", help="extra experiment switches for every run, e.g. OSA_CPS=16,OSA_CPS_LDS=150000")
    ap.add_argument("--split", action="store_true", help="f16x3: split input and split output, as inside the GwcNet aggregation chain (3-D conv layers)")
    ap.add_argument("--envs", default="", help="semicolon list of experiment-switch sets to sweep per layer, e.g. 'OSA_MARCH=0;OSA_MARCH_GEO=0;OSA_MARCH_GEO=1,OSA_MARCH_NSEG=2' (experiments build)")
    ap.add_argument("--dbgs", default="", help="comma list of OSA_DBG masks to sweep (timing-only kernel ablations)")
    args = ap.parse_args()
    dev = "cuda:0"
    for kv in args.env.split(","):
        if "=" in kv:
            os.environ[kv.split("=")[0]] = kv.split("=")[1]
    cfgs = [None] + [int(c) for c in args.cfgs.split(",") if c != ""]
    if args.dbgs:       # ablation sweep: reuse the cfg loop, value = -(mask) - 1
        cfgs = [None] + [-int(c) - 1 for c in args.dbgs.split(",")]
    if args.envs:       # switch-set sweep: reuse the cfg loop, value = the "K=V,K=V" string
        cfgs = [e for e in args.envs.split(";")]
    total = {}
    layers = [l + (1,) for l in LAYERS] if args.set == "3d" else (LAYERS_2D if args.set == "2d" else LAYERS_GRU)
    nb = args.batch * (2 if args.set == "2d" else 1)
    for name, kind, Ci, Co, k, s, dims, count, dil in layers:
        if args.only and args.only not in name:
            continue
        if kind == "conv2d":
            m = nn.Conv2d(Ci, Co, k, s, dil * (k // 2), dilation=dil, bias=False)
        elif kind == "deconv":
            m = nn.ConvTranspose3d(Ci, Co, k, stride=2, padding=1, output_padding=1, bias=False)
        else:
            m = nn.Conv3d(Ci, Co, k, s, k // 2, bias=False)
        m = m.to(dev)
        x = ops.empty_cl(nb, Ci, *dims, dev)
        x.normal_()
        ranges.ensure_meta(x)          # one range reduction, not one per call
        split = args.split and kind == "conv" and Ci % 16 == 0 and Co % 16 == 0
        if split:                      # split image of x through an identity 1x1x1 layer
            idm = nn.Conv3d(Ci, Ci, 1, bias=False).to(dev)
            idm.weight.data = torch.eye(Ci, device=dev).reshape(Ci, Ci, 1, 1, 1).clone()
            x = PackedConv3d(idm, None, 0)(x, out_split=True)
        layer = SmallCoConv3d(m) if kind == "small" else PackedConv3d(m, (nn.BatchNorm2d(Co) if kind == "conv2d" else nn.BatchNorm3d(Co)).to(dev).eval(), 1)
        od = layer.out_shape(*dims) if kind != "small" else dims
        macs = nb * Ci * Co * (k ** (2 if kind == "conv2d" else 3)) * (od[0] * od[1] * od[2]) / (8 if kind == "deconv" else 1)
        line = f"{name:26s} {macs / 1e9:7.2f} GMAC x{count}"
        for cfg in cfgs:
            os.environ.pop("OSA_CONV_CFG", None)
            os.environ.pop("OSA_DBG", None)
            swept = []
            if isinstance(cfg, str):
                for kv in cfg.split(","):
                    if "=" in kv:
                        os.environ[kv.split("=")[0]] = kv.split("=")[1]; swept.append(kv.split("=")[0])
            elif cfg is not None and cfg < 0:
                os.environ["OSA_DBG"] = str(-cfg - 1)
            elif cfg is not None:
                os.environ["OSA_CONV_CFG"] = str(cfg)
            from openstereo_amd import _lib as _L
            if hasattr(_L.load(), "osa_conv_b_ring_mask") and os.environ.get("OSA_B_RING_MASK"):   # a run-time switch of the shipped library (C ABI); unset: its built-in default
                _L.load().osa_conv_b_ring_mask(int(os.environ["OSA_B_RING_MASK"], 0))
            try:
                call = (lambda: layer(x, out_split=True)) if split else (lambda: layer(x))
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    call()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.iters
                line += f" | cfg {cfg if cfg is not None else 'auto'}: {ms:7.3f} ms {2 * macs / ms / 1e9:6.1f} TF"
                if cfg is None:
                    total["auto"] = total.get("auto", 0) + ms * count
            except Exception as ex:  # config not applicable to this layer
                line += f" | cfg {cfg}: n/a ({str(ex)[:40]})"
            for k_ in swept:
                os.environ.pop(k_, None)
        print(line, flush=True)
    os.environ.pop("OSA_CONV_CFG", None)
    os.environ.pop("OSA_DBG", None)
    print(f"sum over one forward (auto cfg): {total.get('auto', 0):.3f} ms")


if __name__ == "__main__":
    main()
