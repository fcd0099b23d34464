"""Diagnostic (GPU): split-output f16x3 layers, shipped library (transposed-accumulator epilogue) vs a -DOSA_XT=0 build, bit for bit.
    python tools/diag_xt.py            (parent: runs itself twice with OSA_LIB_PATH set / unset and compares the dumps)"""
import os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..")
CASES = [("32->32 k3 12x34x60", 32, 32, 3, (12, 34, 60), 1), ("32->32 k3 48x136x240 (cfg 0)", 32, 32, 3, (48, 136, 240), 1),
         ("64->64 k3 24x136x240 (cfg 1)", 64, 64, 3, (24, 136, 240), 1), ("128->128 k3 12x136x240 (cfg 2)", 128, 128, 3, (12, 136, 240), 1),
         ("64->32 k3 48x136x240", 64, 32, 3, (48, 136, 240), 1), ("32->64 s2 48x136x240", 32, 64, 3, (48, 136, 240), 2),
         ("2-D 128->128 272x480 x4 (cfg 9)", 128, 128, 3, (1, 272, 480), 1)]

def child(tag):
    import torch, torch.nn as nn
    sys.path.insert(0, ROOT)
    from openstereo_amd import ops, engine
    from openstereo_amd.engine import PackedConv3d, ACT_RELU
    engine.set_precision("f16x3")
    torch.manual_seed(0)
    for ci, (name, Ci, Co, k, dims, s) in enumerate(CASES):
        B = 4 if dims[0] == 1 else 1
        conv = nn.Conv3d(Ci, Co, k, s, k // 2, bias=False).cuda()
        bn = nn.BatchNorm3d(Co).cuda().eval()
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2); bn.weight.data.normal_(1, 0.2); bn.bias.data.normal_()
        if dims[0] == 1:
            c2 = nn.Conv2d(Ci, Co, k, s, k // 2, bias=False).cuda(); c2.weight.data = conv.weight.data[:, :, 1].contiguous() if k == 3 else conv.weight.data[:, :, 0]
            layer = PackedConv3d(c2, nn.BatchNorm2d(Co).cuda().eval(), ACT_RELU)
        else:
            layer = PackedConv3d(conv, bn, ACT_RELU)
        x = ops.to_cl(torch.randn(B, Ci, *dims, device="cuda"))
        y = layer(x, out_split=True)
        outs = [y]
        if s == 1 and Ci == Co:
            outs.append(layer(x, residual=y, out_split=True))                     # split residual
        torch.cuda.synchronize()
        for j, o in enumerate(outs):
            raw = o.permute(0, 2, 3, 4, 1).contiguous().cpu().numpy().view(np.uint32)     # [B,D,H,W,C] raw words of the split layout
            np.save(f"/tmp/xt_{tag}_{ci}_{j}.npy", raw)

if len(sys.argv) > 1:
    child(sys.argv[1]); sys.exit(0)
env = dict(os.environ)
subprocess.run([sys.executable, __file__, "xt"], env=env, check=True)
env["OSA_LIB_PATH"] = os.path.join(ROOT, "openstereo_amd/lib/variants/noxt.so")
subprocess.run([sys.executable, __file__, "ref"], env=env, check=True)
for ci, (name, *_rest) in enumerate(CASES):
    for j in range(2):
        fa, fb = f"/tmp/xt_xt_{ci}_{j}.npy", f"/tmp/xt_ref_{ci}_{j}.npy"
        if not os.path.exists(fa):
            continue
        a, b = np.load(fa), np.load(fb)
        def dec(w):                      # split layout: per 16-channel block 8 words of hi halves, 8 words of lo halves -> hi + lo (scaled values)
            blk = w.reshape(*w.shape[:-1], -1, 16)
            hi = blk[..., :8].copy().view(np.float16).astype(np.float64); lo = blk[..., 8:].copy().view(np.float16).astype(np.float64)
            return (hi + lo).reshape(*w.shape[:-1], -1)
        va, vb = dec(a), dec(b)
        scale = np.abs(vb).max()
        err = np.abs(va - vb) / scale
        bad = err > 1e-5
        print(f"   decoded: max |diff| / max |ref| = {err.max():.2e}; elements off by > 1e-5: {int(bad.sum())} of {bad.size}"
              + (f"; channels {sorted(set(np.argwhere(bad)[:, 4].tolist()))[:40]}; w {sorted(set(np.argwhere(bad)[:, 3].tolist()))[:24]}; h {sorted(set(np.argwhere(bad)[:, 2].tolist()))[:12]}; sample xt {va[bad][:4]} ref {vb[bad][:4]}" if bad.any() else ""))
        d = a != b
        msg = f"{name:36s} {'with split residual' if j else 'plain':20s}: {int(d.sum())} of {d.size} words differ"
        if d.any():
            idx = np.argwhere(d)
            pass
        print(msg)
