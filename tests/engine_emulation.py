"""TEST INFRASTRUCTURE: a torch-CPU stand-in for the engine's layer classes and entry points.

The engine forwards in `openstereo_amd.models` are wiring: which conv gets which BatchNorm, which residual, which
channel slice of which buffer, in which order.  The kernels behind `PackedConv3d` & co. are pinned on the GPU
(`tests/test_gpu_parity.py`); this module lets the SAME forward code run on the CPU with every engine layer replaced
by the torch operator it stands for, so that the grafts of `openstereo_amd.attach.patch_reference_modules()` can be
checked against the REAL reference classes in the build container (where there is no GPU) -- reference-built model,
engine forward, reference parameters, compared with the reference's own forward.

Never imported by the product.  `install()` returns an `uninstall` callable."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_RELU6, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4, 5


def empty_cl(B, C, D, H, W, device, dtype=torch.float32):
    return torch.zeros((B, D, H, W, C), device=device, dtype=dtype).permute(0, 4, 1, 2, 3)


def to_cl(x, pad_to=4):
    B, C, D, H, W = x.shape
    Cp = (C + pad_to - 1) // pad_to * pad_to
    y = empty_cl(B, Cp, D, H, W, x.device)
    y[:, :C] = x.float()
    return y


def to_ncdhw(x, channels=None):
    return x[:, :(x.shape[1] if channels is None else channels)].contiguous()


def _act(y, act, slope):
    return {ACT_NONE: lambda t: t, ACT_RELU: F.relu, ACT_LEAKY: lambda t: F.leaky_relu(t, slope), ACT_RELU6: F.relu6,
            ACT_SIGMOID: torch.sigmoid, ACT_TANH: torch.tanh}[act](y)


class PackedConv3d:
    """conv / transposed conv + eval BatchNorm + residual / fused redir + activation + gate, channel-slice I/O --
    the call contract of openstereo_amd.engine.PackedConv3d, computed with torch CPU operators."""

    def __init__(self, conv, bn=None, act=ACT_NONE, slope=0.01, precision=None):
        self.conv, self.bn, self.act, self.slope, self.precision = conv, bn, act, float(slope), precision or "f32"
        self.flat = isinstance(conv, (nn.Conv2d, nn.ConvTranspose2d))
        self.transposed = isinstance(conv, (nn.ConvTranspose3d, nn.ConvTranspose2d))
        self.Ci, self.Co = (conv.in_channels, conv.out_channels)
        t3 = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)
        self.k = ((1,) + tuple(conv.kernel_size)) if self.flat else t3(conv.kernel_size)

    def _raw(self, xin):
        if self.flat:
            return self.conv(xin[:, :, 0]).unsqueeze(2)
        return self.conv(xin)

    def out_shape(self, D, H, W):
        with torch.no_grad():
            return tuple(self._raw(torch.zeros(1, self.Ci, D, H, W)).shape[2:])

    def __call__(self, x, residual=None, out=None, gate=None, x_off=0, out_off=0, res_off=0, gate_raw=False, redir=None,
                 out_split=False):
        assert x.shape[1] >= x_off + self.Ci, "input slice out of range"
        with torch.no_grad():
            y = self._raw(x[:, x_off:x_off + self.Ci])
            if self.bn is not None:
                assert not self.bn.training
                y = (self.bn(y[:, :, 0]).unsqueeze(2) if self.flat else self.bn(y))
            if residual is not None:
                y = y + residual[:, res_off:res_off + self.Co]
            if redir is not None:
                rl, rt = redir
                assert rl.act == ACT_NONE and residual is None
                y = y + rl(rt)[:, :self.Co]
            y = _act(y, self.act, self.slope)
            if gate is not None:
                g = gate[..., :self.Co].permute(0, 3, 1, 2).unsqueeze(2)
                y = y * (g if gate_raw else torch.sigmoid(g))
            if out is None:
                out = empty_cl(y.shape[0], (self.Co + 3) // 4 * 4, *y.shape[2:], y.device)
            assert tuple(out.shape[2:]) == tuple(y.shape[2:])
            out[:, out_off:out_off + self.Co] = y
        return out


class SmallCoConv3d:
    def __init__(self, conv):
        self.conv, self.Ci, self.Co = conv, conv.in_channels, conv.out_channels

    def __call__(self, x, residual=None):
        with torch.no_grad():
            y = self.conv(x[:, :self.Ci])
            return y if residual is None else y + residual


class DepthwiseConv2d:
    def __init__(self, conv, bn=None, act=ACT_NONE):
        self.conv, self.bn, self.act, self.C = conv, bn, act, conv.in_channels

    def __call__(self, x, add=None):
        with torch.no_grad():
            y = self.conv(x[:, :self.C, 0])
            if self.bn is not None:
                y = self.bn(y)
            y = _act(y, self.act, 0.01).unsqueeze(2)
            if add is not None:
                y = y + add[:, :self.C]
            out = empty_cl(y.shape[0], self.C, 1, *y.shape[3:], y.device)
            out[:] = y
        return out


# ---- entry points of openstereo_amd.ops the forwards call (oracle arithmetic) --------------------------------------
def _vol(lg, rg, G, lc, rc, maxdisp, mask_left=True):
    from oracle import torch_ref as O
    parts = []
    if lg is not None:
        parts.append(O.gwc_volume(lg.float(), rg.float(), maxdisp, G))
    if lc is not None:
        parts.append(O.concat_volume(lc.float(), rc.float(), maxdisp, mask_left))
    return to_cl(torch.cat(parts, 1))


def build_cost_volume_cl(gwc_left, gwc_right, num_groups, cat_left=None, cat_right=None, maxdisp=48, mask_left=True):
    return _vol(gwc_left, gwc_right, num_groups, cat_left, cat_right, maxdisp, mask_left)


def build_cost_volume_from_cl(gwc_feat, num_groups, cat_feat, B, maxdisp, gwc_channels=None, cat_channels=None, gwc_off=0,
                              mask_left=True, out_split=False):          # (out_split: a storage format of the GPU engine, same values)
    C = gwc_feat.shape[1] - gwc_off if gwc_channels is None else gwc_channels
    g = gwc_feat[:, gwc_off:gwc_off + C, 0]
    c = None if cat_feat is None else cat_feat[:, :(cat_feat.shape[1] if cat_channels is None else cat_channels), 0]
    return _vol(g[:B], g[B:], num_groups, None if c is None else c[:B], None if c is None else c[B:], maxdisp, mask_left)


def upsample_softargmin(cost_lowres, maxdisp, h, w, align_corners=False):
    from oracle import torch_ref as O
    c = cost_lowres if cost_lowres.dim() == 5 else cost_lowres[:, None]
    return O.upsample_regression(c.contiguous(), maxdisp, h, w, align_corners)


def softmax_disparity_regression(cost, maxdisp=None, keepdim=True, return_prob=False):
    from oracle import torch_ref as O
    out = O.softmax_regression(cost, keepdim)
    return (out, F.softmax(cost, 1)) if return_prob else out


def install():
    """Swap the engine layer classes / entry points inside the model modules for the emulation. -> uninstall()"""
    from openstereo_amd import ops
    from openstereo_amd.models import gwcnet, psmnet, igev_style, lightstereo, igev_update
    saved = []

    def swap(obj, name, new):
        saved.append((obj, name, getattr(obj, name)))
        setattr(obj, name, new)
    for mod in (gwcnet, psmnet, igev_style, lightstereo, igev_update):
        for name, new in (("PackedConv3d", PackedConv3d), ("SmallCoConv3d", SmallCoConv3d), ("DepthwiseConv2d", DepthwiseConv2d),
                          ("on_engine", lambda t: True)):
            if hasattr(mod, name):
                swap(mod, name, new)
    for name, new in (("on_engine", lambda t: True), ("to_cl", to_cl), ("to_ncdhw", to_ncdhw), ("empty_cl", empty_cl),
                      ("build_cost_volume_cl", build_cost_volume_cl), ("build_cost_volume_from_cl", build_cost_volume_from_cl),
                      ("upsample_softargmin", upsample_softargmin), ("softmax_disparity_regression", softmax_disparity_regression)):
        swap(ops, name, new)

    def uninstall():
        while saved:
            obj, name, old = saved.pop()
            setattr(obj, name, old)
    return uninstall
