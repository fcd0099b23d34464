"""Diagnostic (GPU): channel / voxel mapping of the transposed-accumulator epilogue, on layers whose outputs are known in closed form."""
import os, sys
import numpy as np, torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openstereo_amd import ops, engine
from openstereo_amd.engine import PackedConv3d, ACT_NONE
engine.set_precision("f16x3")

def dec(o, scale_ref=None):
    w = o.permute(0, 2, 3, 4, 1).contiguous().cpu().numpy().view(np.uint32)
    blk = w.reshape(*w.shape[:-1], -1, 16)
    hi = blk[..., :8].copy().view(np.float16).astype(np.float64); lo = blk[..., 8:].copy().view(np.float16).astype(np.float64)
    return (hi + lo).reshape(*w.shape[:-1], -1)

D, H, W = 4, 8, 8
x = torch.zeros(1, 32, D, H, W)
vox = torch.arange(D * H * W).reshape(D, H, W).float()
for c in range(32):
    x[0, c] = 1000.0 * c + vox                       # value encodes (channel, voxel)
xc = ops.to_cl(x.cuda())
for tag, wmode in (("identity 1x1x1 (out[v][c] = x[v][c])", "eye"), ("zero weights, BN shift = channel index", "shift")):
    conv = nn.Conv3d(32, 32, 1, bias=False).cuda()
    bn = nn.BatchNorm3d(32).cuda().eval()
    bn.running_mean.zero_(); bn.running_var.fill_(1.0 - bn.eps); bn.weight.data.fill_(1.0); bn.bias.data.zero_()
    if wmode == "eye":
        conv.weight.data = torch.eye(32).reshape(32, 32, 1, 1, 1).cuda()
    else:
        conv.weight.data.zero_(); bn.bias.data = torch.arange(32.0).cuda()
    y = PackedConv3d(conv, bn, ACT_NONE)(xc, out_split=True)
    torch.cuda.synchronize()
    v = dec(y)
    scale = v.max() / (31000.0 + D * H * W - 1 if wmode == "eye" else 31.0)
    v = v / scale
    print("==", tag, " (split scale", scale, ")")
    for (d, h, w) in ((0, 0, 0), (0, 0, 1), (0, 1, 0), (1, 0, 0), (3, 7, 7)):
        print(f"   voxel ({d},{h},{w}) = #{d * 64 + h * 8 + w}:", np.round(v[0, d, h, w], 1).tolist())
