"""IGEV / StereoBase iterative-refinement update block on the gfx950 engine (BASELINE
configs[4] "IGEV iterative GRU refinement").

Mirror of stereo/modeling/models/igev/update.py:17-144 (stereobase/gru_blocks.py:233-328 is the same
block with a configurable correlation-plane count): same class names, constructor arguments and
state_dict keys; forward on the engine with NHWC tensors:

  * every Conv2d (3x3, 1x1, 7x7, with bias) is the MFMA conv kernel with D = 1; inputs that the
    reference concatenates (`torch.cat([h, x...])`) are channel slices of one buffer,
  * ConvGRU (r3): every level keeps ONE state buffer [h | x | r*h | z] per forward (`_GruLevel`).  The producers of x -- pool2x /
    interp of the neighbouring levels' hidden states (`osa_pool2x_nhwc_f32`, `osa_resize_bilinear_nhwc_f32`) and the motion encoder's
    last conv -- write their channel slice in place; ONE launch computes convr and convz (they read the same [h | x]: the brick is
    staged once, twice the workgroups on the small 1/8 and 1/16 maps) with the sigmoid, the `+ cr / + cz` and `r * h` (h as raw gate of
    the first half, OSA_GATE_CHANNELS) in its epilogue and stores [r*h | z] next to x; convq reads [x | r*h] in place (its input
    channels permuted when the weights are packed) with tanh / `+ cq` fused; h' = (1-z)*h + z*q (`osa_gru_combine_f32`) updates h in
    place.  No torch.cat, no clone, no copy per call: three launches per GRU.

forward() takes and returns the reference's NCHW tensors; forward_cl() is the channels-last entry.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _ext, _lib, amp
from .. import autograd as AG
from ..engine import cached_pack, PackedConv3d, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH
from ..ops import empty_cl, is_cl, on_engine, _stream
from ..ranges import attach_meta, combine_meta, ensure_meta, fold_amax, inherit_meta, meta_of, new_meta
from .lightstereo import nchw_to_cl, cl_to_nchw


def _nhwc(x):
    """NHWC 4-D view [B,H,W,C] of an engine tensor (logical [B,C,1,H,W])."""
    B, C, _, H, W = x.shape
    return x.permute(0, 2, 3, 4, 1).reshape(B, H, W, C)


def _as_cl(t4):
    """4-D logical NCHW tensor (any strides) -> engine tensor, without a copy when it already is NHWC."""
    B, C, H, W = t4.shape
    if C % 4 == 0 and t4.dtype == torch.float32 and t4.stride() == (H * W * C, 1, W * C, C):
        return t4.unsqueeze(2)
    return nchw_to_cl(t4)


def _fold_meta(dst, src):
    """dst's range block (f16x3 chains) must cover values copied / resampled from src: slot-wise maximum, one tiny kernel."""
    m = meta_of(dst)
    if m is not None:
        torch.maximum(m, ensure_meta(src), out=m)


class _GruLevel:
    """State of one ConvGRU level for the duration of a forward: T = [h (hd) | x (cx) | r*h (hd) | z (hd)] NHWC, plus the static
    context terms: crz = [cr | cz] (residual of the fused r|z launch) and cq."""
    __slots__ = ("T", "hd", "cx", "B", "H", "W", "crz", "cq", "q")

    def __init__(self, h, cx, cz, cr, cq, track):
        B, hd, _, H, W = h.shape
        self.hd, self.cx, self.B, self.H, self.W = hd, cx, B, H, W
        self.T = empty_cl(B, 3 * hd + cx, 1, H, W, h.device)      # the r*h | z slots are written by the first r|z launch before anything reads them
        if track:
            attach_meta(self.T)
        self.set_h(h)
        self.crz = _cat_cl([cr, cz], h.device)
        self.cq = cq
        self.q = empty_cl(B, hd, 1, H, W, h.device)

    def set_h(self, h):
        self.T[:, :self.hd] = h[:, :self.hd]
        _fold_meta(self.T, h)

    def write_x(self, off, part):
        """copy an engine tensor into x channels [off, off + C) (standalone ConvGRU calls; the update block's producers write in place)"""
        C = part.shape[1]
        self.T[:, self.hd + off:self.hd + off + C] = part
        _fold_meta(self.T, part)

    def h_out(self):
        return self.T                                   # consumers read channels [0, hd) (PackedConv3d takes Cs >= Ci; cl_to_nchw slices)


def _resample_into(kind, src, C, dst, off):
    """pool2x (update.py:99-100) or interp (:107-109, bilinear align_corners=True) of channels [0, C) of `src` into channels [off, off + C) of
    the level buffer `dst`, both NHWC engine tensors; the destination's range block inherits the source's maximum in-kernel."""
    B, sCs, _, H, W = src.shape
    _, dCs, _, Hd, Wd = dst.shape
    sm, dm = meta_of(src), meta_of(dst)
    if sm is None or dm is None:
        sm = dm = None
    if kind == "pool":
        assert (Hd, Wd) == ((H - 1) // 2 + 1, (W - 1) // 2 + 1), ((H, W), (Hd, Wd))
    ext = _ext.load()
    if ext is not None:
        ext.resample_nhwc(src, dst, off, 0 if kind == "pool" else 1, [B, H, W, C, sCs, dCs] if kind == "pool" else [B, H, W, Hd, Wd, C, sCs, dCs], sm, dm)
        return
    xm, ym = (None, None) if sm is None else (sm.data_ptr(), dm.data_ptr())
    if kind == "pool":
        _lib.call("osa_pool2x_nhwc_f32", src.data_ptr(), dst.data_ptr() + 4 * off, B, H, W, C, sCs, dCs, xm, ym, _stream())
    else:
        _lib.call("osa_resize_bilinear_nhwc_f32", src.data_ptr(), dst.data_ptr() + 4 * off, B, H, W, Hd, Wd, C, sCs, dCs, xm, ym, _stream())


def pool2x(x):
    """update.py:99-100 on an engine tensor -> new engine tensor (averages: x's range block stays valid)."""
    B, C, _, H, W = x.shape
    out = inherit_meta(empty_cl(B, C, 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1, x.device), x)
    _resample_into("pool", x, C, out, 0)
    return out


def interp(x, dest):
    """update.py:107-109 (bilinear, align_corners=True) on engine tensors (convex combinations: x's range block stays valid)."""
    B, C, _, _, _ = x.shape
    out = inherit_meta(empty_cl(B, C, 1, dest.shape[3], dest.shape[4], x.device), x)
    _resample_into("interp", x, C, out, 0)
    return out


def _cat_cl(parts, dev, track=False):
    """Channel concatenation into one NHWC buffer (what torch.cat does for the reference).  track (f16x3 chains): the result
    gets a range block = slot-wise maximum of the parts' blocks; a part that has none yet is measured once and keeps its block."""
    B, _, _, H, W = parts[0].shape
    C = sum(p.shape[1] for p in parts)
    out = empty_cl(B, C, 1, H, W, dev)
    o = 0
    for p in parts:
        assert p.shape[1] % 4 == 0, "engine tensors are channel-padded to 4: a part with padded channels would shift the next one"
        out[:, o:o + p.shape[1]] = p
        o += p.shape[1]
    if track:
        out._osa_meta = combine_meta(*[ensure_meta(p) for p in parts])
    return out


class DispHead(nn.Module):
    """update.py:17-26"""

    def __init__(self, input_dim=128, hidden_dim=256, output_dim=1):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, output_dim, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self._eng = None

    def forward_cl(self, x):
        e = cached_pack(self, "_eng", lambda: (PackedConv3d(self.conv1, None, ACT_RELU), PackedConv3d(self.conv2)))
        return e[1](e[0](x))

    @amp.contract("cast")
    def forward(self, x):
        if self.training or (torch.is_grad_enabled() and x.requires_grad):       # update.py:25-26, convs on the engine with autograd
            with AG.engine_convs():
                return self.conv2(self.relu(self.conv1(x)))
        return cl_to_nchw(self.forward_cl(nchw_to_cl(x)), self.conv2.out_channels)


FUSED_GRU_TRAIN = os.environ.get("OSA_FUSED_GRU_TRAIN", "1") != "0"      # training: paired r|z conv + fused gate kernels (ConvGRU.forward_train)


class ConvGRU(nn.Module):
    """update.py:29-45"""

    def __init__(self, hidden_dim, input_dim, kernel_size=3):
        super().__init__()
        k, p = kernel_size, kernel_size // 2
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, k, padding=p)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, k, padding=p)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, k, padding=p)
        self.hidden_dim = hidden_dim
        self._eng = None

    def _packs(self):
        """(fused r|z launch, q launch): convr and convz stacked on the output axis (update.py:38-39 read the same hx); convq with its input
        channels rotated from [r*h | x] to [x | r*h], the order in which the level buffer holds them."""
        def build():
            hd = self.convz.out_channels
            k, pd = self.convz.kernel_size, self.convz.padding
            with torch.no_grad():
                rz = nn.Conv2d(self.convz.in_channels, 2 * hd, k, padding=pd).to(self.convz.weight.device)
                rz.weight.copy_(torch.cat([self.convr.weight, self.convz.weight], 0)); rz.bias.copy_(torch.cat([self.convr.bias, self.convz.bias], 0))
                q = nn.Conv2d(self.convq.in_channels, hd, k, padding=pd).to(self.convq.weight.device)
                q.weight.copy_(torch.cat([self.convq.weight[:, hd:], self.convq.weight[:, :hd]], 1)); q.bias.copy_(self.convq.bias)
            return PackedConv3d(rz, None, ACT_SIGMOID), PackedConv3d(q, None, ACT_TANH)
        return cached_pack(self, "_eng", build)

    def new_level(self, h, cz, cr, cq, cx):
        prz, _ = self._packs()
        assert self.convz.out_channels % 4 == 0 and h.shape[1] >= self.convz.out_channels and cx % 4 == 0
        return _GruLevel(h, cx, cz, cr, cq, track=prz.precision == "f16x3")

    def step(self, lv):
        """One GRU update of a level whose x slots are filled: 3 launches, h updated in place (update.py:36-45)."""
        prz, pq = self._packs()
        hd, cx, T = lv.hd, lv.cx, lv.T
        assert prz.Ci == hd + cx, (prz.Ci, hd, cx)
        prz(T, residual=lv.crz, gate=_nhwc(T), gate_raw=True, gate_channels=hd, out=T, out_off=hd + cx)    # [sigmoid(convr+cr)*h | sigmoid(convz+cz)]
        pq(T, x_off=hd, residual=lv.cq, out=lv.q)                                                           # tanh(convq([r*h, x]) + cq)
        ext = _ext.load()
        if ext is not None:
            ext.gru_combine(T, 2 * hd + cx, lv.q, T, T, [lv.B * lv.H * lv.W, hd, T.shape[1], lv.q.shape[1], T.shape[1], T.shape[1]], meta_of(T))
            return
        zoff = 4 * (2 * hd + cx)
        _lib.call("osa_gru_combine_f32", T.data_ptr() + zoff, lv.q.data_ptr(), T.data_ptr(), T.data_ptr(), lv.B * lv.H * lv.W, hd,
                  T.shape[1], lv.q.shape[1], T.shape[1], T.shape[1], None if meta_of(T) is None else meta_of(T).data_ptr(), _stream())

    def forward_cl(self, h, cz, cr, cq, *x_list):
        """Stand-alone call on engine tensors (the update block drives `step` on persistent levels instead)."""
        lv = self.new_level(h, cz, cr, cq, sum(x.shape[1] for x in x_list))
        off = 0
        for x in x_list:
            lv.write_x(off, x)
            off += x.shape[1]
        self.step(lv)
        out = empty_cl(*h.shape, h.device)
        out[:, :lv.hd] = lv.T[:, :lv.hd]
        if h.shape[1] > lv.hd:
            out[:, lv.hd:] = 0.0
        return inherit_meta(out, lv.T)

    def forward_train(self, h, cz, cr, cq, *x_list):
        """update.py:36-45 in training mode.  r5: convz | convr as ONE engine layer (they read the same [h | x]: one forward, one data
        gradient, one weight gradient), convq as another, and the gate arithmetic in two fused kernels forward and two backward
        (csrc/gru_train.hip) instead of ~13 + ~20 torch elementwise launches per cell -- 66 cells per StereoBase training step.
        FUSED_GRU_TRAIN = False (or operands the fused form does not cover) keeps the torch composition."""
        hd = self.hidden_dim
        x = torch.cat(x_list, dim=1) if len(x_list) > 1 else x_list[0]
        fused = FUSED_GRU_TRAIN and hd % 4 == 0 and h.is_cuda and all(t.dtype in (torch.float16, torch.float32) for t in (h, cz, cr, cq, x)) \
            and tuple(self.convz.stride) == (1, 1) and tuple(self.convz.dilation) == (1, 1) and (h.shape[1] + x.shape[1]) % 4 == 0
        if not fused:
            with AG.engine_convs():
                hx = torch.cat([h, x], dim=1)
                z = torch.sigmoid(self.convz(hx) + cz)
                r = torch.sigmoid(self.convr(hx) + cr)
                q = torch.tanh(self.convq(torch.cat([r * h, x], dim=1)) + cq)
            return (1 - z) * h + z * q
        cdt = amp.conv_out_dtype(h)                                   # dtype the reference's convolutions return here (autocast: fp16)
        pad = self.convz.padding
        hx = torch.cat([h, x], dim=1)
        pre = AG.conv2d_pair(hx, self.convz.weight, self.convr.weight, padding=pad)                   # [convz(hx) | convr(hx)], fp32, no bias
        z, rh = AG.gru_gates_rz(pre, self.convz.bias, self.convr.bias, cz, cr, h,
                                rh_dtype=torch.promote_types(torch.promote_types(cdt, cr.dtype), h.dtype))   # r * h as the reference rounds it
        qpre = AG.conv2d(torch.cat([rh, x], dim=1), self.convq.weight, None, 1, pad)
        out_dt = torch.promote_types(torch.promote_types(torch.promote_types(cdt, cz.dtype), cq.dtype), h.dtype)
        return AG.gru_gates_q(z, qpre, self.convq.bias, cq, h, out_dtype=out_dt)

    @amp.contract("gru")
    def forward(self, h, cz, cr, cq, *x_list):
        if self.training or (torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self.forward_train(h, cz, cr, cq, *x_list)
        c = nchw_to_cl
        return cl_to_nchw(self.forward_cl(c(h), c(cz), c(cr), c(cq), *[c(x) for x in x_list]), self.convz.out_channels)


class BasicMotionEncoder(nn.Module):
    """update.py:72-92; `cor_planes` overrides the IGEV formula for StereoBase (gru_blocks.py:236)."""

    def __init__(self, args, cor_planes=None):
        super().__init__()
        self.args = args
        if cor_planes is None:
            cor_planes = args.CORR_LEVELS * (2 * args.CORR_RADIUS + 1) * (8 + 1)
        self.convc1 = nn.Conv2d(cor_planes, 64, 1, padding=0)
        self.convc2 = nn.Conv2d(64, 64, 3, padding=1)
        self.convd1 = nn.Conv2d(1, 64, 7, padding=3)
        self.convd2 = nn.Conv2d(64, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 64, 128 - 1, 3, padding=1)
        self._eng = None

    def forward_cl(self, disp, corr, out=None, out_off=0, disp_in_place=False):
        """disp: engine tensor with the disparity in channel 0 (channels 1..3 zero); corr: engine tensor.  out / out_off: write the 128
        motion-feature channels into a channel slice of `out` (the 1/4 GRU level's x slot) instead of a new tensor; disp_in_place: the
        caller has already written the disparity into channel out_off + 127 (osa_disp_update_f32)."""
        R = lambda m: PackedConv3d(m, None, ACT_RELU)
        e = cached_pack(self, "_eng", lambda: dict(c1=R(self.convc1), c2=R(self.convc2), d1=R(self.convd1), d2=R(self.convd2),
                                                   conv=R(self.conv)))
        B, _, _, H, W = disp.shape
        cor_disp = empty_cl(B, 128, 1, H, W, disp.device)      # [cor | disp_]
        e["c2"](e["c1"](corr), out=cor_disp, out_off=0)
        e["d2"](e["d1"](disp), out=cor_disp, out_off=64)
        if out is None:
            out = empty_cl(B, 128, 1, H, W, disp.device)       # [conv(cor_disp) (127) | disp]
        e["conv"](cor_disp, out=out, out_off=out_off)
        if not disp_in_place:
            out[:, out_off + 127] = disp[:, 0]
            if meta_of(out) is not None:
                fold_amax(out, disp[:, 0])
        return out

    def forward_train(self, disp, corr):
        """update.py:83-92 (the 7x7 conv on the 1-channel disparity stays a torch op: fewer than 4 input channels)"""
        with AG.engine_convs():
            cor = F.relu(self.convc2(F.relu(self.convc1(corr))))
            d = F.relu(self.convd2(F.relu(self.convd1(disp))))
            out = F.relu(self.conv(torch.cat([cor, d], dim=1)))
        return torch.cat([out, disp], dim=1)

    @amp.contract("enc")
    def forward(self, disp, corr):
        if self.training or (torch.is_grad_enabled() and (disp.requires_grad or corr.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self.forward_train(disp, corr)
        return cl_to_nchw(self.forward_cl(nchw_to_cl(disp), nchw_to_cl(corr)), 128)


class BasicMultiUpdateBlock(nn.Module):
    """update.py:112-144"""

    def __init__(self, args, hidden_dims=[], cor_planes=None):
        super().__init__()
        self.args = args
        self.encoder = BasicMotionEncoder(args, cor_planes)
        encoder_output_dim = 128
        self.gru04 = ConvGRU(hidden_dims[2], encoder_output_dim + hidden_dims[1] * (args.N_GRU_LAYERS > 1))
        self.gru08 = ConvGRU(hidden_dims[1], hidden_dims[0] * (args.N_GRU_LAYERS == 3) + hidden_dims[2])
        self.gru16 = ConvGRU(hidden_dims[0], hidden_dims[1])
        self.disp_head = DispHead(hidden_dims[2], hidden_dim=256, output_dim=1)
        self.mask_feat_4 = nn.Sequential(nn.Conv2d(hidden_dims[2], 32, 3, padding=1), nn.ReLU(inplace=True))
        self._mask = None

    def reset_engine(self):
        self._mask = None
        self.__dict__.pop("_lv", None)
        for m in self.modules():
            if hasattr(m, "_eng"):
                m._eng = None

    def end_forward(self):
        """Drop the per-level state buffers (and the references to the context tensors they were keyed on).  run_refinement calls this
        when its loop is done; direct forward_cl callers call it between forwards when they rewrite `inp` through raw pointers."""
        self.__dict__.pop("_lv", None)

    def _levels(self, net, inp, n_gru):
        """Per-level state buffers, kept across the calls of one forward: `net[i]` handed back by the previous call IS level i's buffer
        (identity), anything else (first call, a caller that replaced a hidden state) is copied in.  A new set of context tensors
        (`inp`) starts a new forward."""
        ctx = [t for ts in inp[:n_gru] for t in ts]     # the state holds these references, so identity is a safe key (no id() reuse)
        shapes = [tuple(t.shape[2:]) for t in net[:n_gru]]
        # The levels hold a COPY of the context terms (crz = [cr | cz]), so the key also carries what tells a changed content apart: the
        # tensors' version counters (in-place torch updates of static graph inputs bump them) and the capture state (levels made in an eager
        # warm-up must not be reused inside a capture: the crz copy would not be part of the graph).  Writes through raw pointers / `.data`
        # are invisible to both -- call `end_forward()` (or `reset_engine()`) after such an update.
        stamp = ([t._version for t in ctx], torch.cuda.is_current_stream_capturing() if ctx and ctx[0].is_cuda else False)
        st = self.__dict__.get("_lv")
        if st is None or st[0][1] != shapes or st[0][2] != stamp or len(st[0][0]) != len(ctx) or any(a is not b for a, b in zip(st[0][0], ctx)):
            key = (ctx, shapes, stamp)
            hd = [g.convz.out_channels for g in (self.gru04, self.gru08, self.gru16)]
            cx = [128 + (hd[1] if n_gru > 1 else 0), hd[0] + (hd[2] if n_gru > 2 else 0), hd[1]]
            lv = [g.new_level(net[i], inp[i][0], inp[i][1], inp[i][2], cx[i]) if i < n_gru else None
                  for i, g in enumerate((self.gru04, self.gru08, self.gru16))]
            st = (key, lv)
            object.__setattr__(self, "_lv", st)
            return st[1]
        for i, lv in enumerate(st[1]):
            if lv is not None and net[i] is not lv.T:
                lv.set_h(net[i])
        return st[1]

    def forward_cl(self, net, inp, corr=None, disp=None, iter04=True, iter08=True, iter16=True, update=True, want_mask=True, disp_in_place=False):
        """Engine tensors everywhere.  Returns `net` as the per-level state buffers (channels [0, hidden) are the hidden state): hand them
        back unchanged for the next call, or read them with cl_to_nchw(t, hidden).  The levels are cached across the calls of one forward,
        keyed on the identity, version counters and capture state of the `inp` tensors: `inp` must not be rewritten through raw pointers
        between calls without an `end_forward()` in between.  want_mask=False skips mask_feat_4 (returns None for it):
        in test mode only the last iteration's mask features are used (igev_stereo.py:203-207)."""
        n_gru = self.args.N_GRU_LAYERS if hasattr(self, "args") else self.n_gru_layers    # igev/update.py vs stereobase/gru_blocks.py
        lv = self._levels(net, inp, n_gru)
        l4, l8, l16 = lv[0], (lv[1] if n_gru > 1 else None), (lv[2] if n_gru > 2 else None)
        if iter16:
            _resample_into("pool", l8.T, l8.hd, l16.T, l16.hd)                         # x = pool2x(net[1])
            self.gru16.step(l16)
        if iter08:
            _resample_into("pool", l4.T, l4.hd, l8.T, l8.hd)                           # x = [pool2x(net[0]) | interp(net[2], net[1])]
            if n_gru > 2:
                _resample_into("interp", l16.T, l16.hd, l8.T, l8.hd + l4.hd)
            self.gru08.step(l8)
        if iter04:
            self.encoder.forward_cl(disp, corr, out=l4.T, out_off=l4.hd, disp_in_place=disp_in_place)   # x = [motion features | interp(net[1], net[0])]
            if n_gru > 1:
                _resample_into("interp", l8.T, l8.hd, l4.T, l4.hd + 128)
            self.gru04.step(l4)
        out_net = [l.T for l in lv if l is not None] + list(net[n_gru:])
        if not update:
            return out_net
        delta_disp = self.disp_head.forward_cl(l4.T)
        if not want_mask:
            return out_net, None, delta_disp
        mask = cached_pack(self, "_mask", lambda: PackedConv3d(self.mask_feat_4[0], None, ACT_RELU), mods=(self.mask_feat_4,))
        return out_net, mask(l4.T), delta_disp

    def forward_train(self, net, inp, corr=None, disp=None, iter04=True, iter08=True, iter16=True, update=True):
        """update.py:129-150 with differentiable sub-modules (their training paths); pool2x / interp are the reference's torch ops."""
        n_gru = self.args.N_GRU_LAYERS if hasattr(self, "args") else self.n_gru_layers
        # .contiguous(): PyTorch 2.10 + ROCm 7.0 computes a WRONG avg_pool2d gradient for a channels-last input (aten.avg_pool2d_backward
        # with NHWC `self`: 0.9 of max |grad| off vs CPU, tools/diag_cl_ops2.py; the forward is right).  The engine's conv outputs are
        # channels-last, so the hidden states arriving here are too; found by pinning the whole-model training gradients to the reference's
        # CPU autograd (tests/test_gpu_models_e2e.py::test_training_step_matches_reference_autograd: 2-5 % error upstream of the GRUs).
        p2 = lambda t: F.avg_pool2d(t.contiguous(), 3, stride=2, padding=1)
        ip = lambda t, dest: F.interpolate(t, dest.shape[2:], mode="bilinear", align_corners=True)
        net = list(net)
        if iter16:
            net[2] = self.gru16(net[2], *(inp[2]), p2(net[1]))
        if iter08:
            net[1] = self.gru08(net[1], *(inp[1]), p2(net[0]), ip(net[2], net[1])) if n_gru > 2 else self.gru08(net[1], *(inp[1]), p2(net[0]))
        if iter04:
            mf = self.encoder(disp, corr)
            net[0] = self.gru04(net[0], *(inp[0]), mf, ip(net[1], net[0])) if n_gru > 1 else self.gru04(net[0], *(inp[0]), mf)
        if not update:
            return net
        with AG.engine_convs():
            return net, self.mask_feat_4(net[0]), self.disp_head(net[0])

    @amp.contract("update")
    def forward(self, net, inp, corr=None, disp=None, iter04=True, iter08=True, iter16=True, update=True):
        if not on_engine(net[0]):
            raise RuntimeError("openstereo_amd BasicMultiUpdateBlock runs on the GPU engine only (no CPU path)")
        if self.training or (torch.is_grad_enabled() and (net[0].requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self.forward_train(net, inp, corr, disp, iter04, iter08, iter16, update)
        c = nchw_to_cl
        net_cl = [c(t) for t in net]
        inp_cl = [[c(t) for t in ts] for ts in inp]
        res = self.forward_cl(net_cl, inp_cl, None if corr is None else c(corr), None if disp is None else c(disp),
                              iter04, iter08, iter16, update)
        back = lambda lst: [cl_to_nchw(t, r.shape[1]) for t, r in zip(lst, net)]
        if not update:
            return back(res)
        n, mask, delta = res
        return back(n), cl_to_nchw(mask, 32), cl_to_nchw(delta, 1)


class IGEVRefiner(nn.Module):
    """The GRU refinement loop of igev_stereo.py:181-203 (test mode) as one engine module, with the
    reference's attribute name `update_block`:

        geo_fn = Combined_Geo_Encoding_Volume(match_left, match_right, geo_encoding_volume)     (a5, engine)
        for itr in range(iters):
            geo_feat = geo_fn(disp, coords)                                                      (fused lookup kernel)
            [slow-fast schedule: low-res GRUs only]                                              (update block, engine)
            net_list, mask_feat_4, delta_disp = update_block(net_list, inp_list, geo_feat, disp)
            disp = disp + delta_disp
    Hidden states stay NHWC engine tensors across iterations.  Returns the quarter-resolution disparity and
    the mask features that `upsample_disp` consumes."""

    def __init__(self, args, hidden_dims, cor_planes=None):
        super().__init__()
        self.args = args
        self.update_block = BasicMultiUpdateBlock(args, hidden_dims=hidden_dims, cor_planes=cor_planes)

    def forward(self, match_left, match_right, geo_encoding_volume, net_list, inp_list, init_disp, iters):
        return run_refinement(self.update_block, self.args, match_left, match_right, geo_encoding_volume, net_list, inp_list, init_disp, iters)


def run_refinement(update_block, a, match_left, match_right, geo_encoding_volume, net_list, inp_list, init_disp, iters):
    """The loop of IGEVRefiner for any owner of an engine `update_block` (the end-to-end classes of stereo_models.py)."""
    from ..geometry import CombinedGeoEncodingVolume
    if not on_engine(match_left):
        raise RuntimeError("openstereo_amd IGEVRefiner runs on the GPU engine only (no CPU path)")
    geo_fn = CombinedGeoEncodingVolume(match_left.float(), match_right.float(), geo_encoding_volume.float(),
                                       radius=a.CORR_RADIUS, num_levels=a.CORR_LEVELS)
    b, _, h, w = match_left.shape
    coords = torch.arange(w, device=match_left.device).float().reshape(1, 1, w, 1).repeat(b, h, 1, 1)
    wants_grad = torch.is_grad_enabled() and (update_block.training or any(
        t.requires_grad for t in (match_left, match_right, geo_encoding_volume, init_disp, *net_list, *[x for ts in inp_list for x in ts])))
    if wants_grad:
        # training mode / gradient-requiring inputs: the differentiable loop (geometry lookup with its backward kernel, update block
        # through its forward_train path) -- the non-recording forward_cl kernels below would hand back tensors with no graph and the
        # update block would silently receive no gradients.  igev_stereo.py:181-203: disp is detached every iteration.
        net, disp, mask = list(net_list), init_disp.float(), None
        for _ in range(iters):
            disp = disp.detach()
            geo_feat = geo_fn(disp, coords)
            if a.N_GRU_LAYERS == 3 and a.SLOW_FAST_GRU:
                net = update_block(net, inp_list, iter16=True, iter08=False, iter04=False, update=False)
            if a.N_GRU_LAYERS >= 2 and a.SLOW_FAST_GRU:
                net = update_block(net, inp_list, iter16=a.N_GRU_LAYERS == 3, iter08=True, iter04=False, update=False)
            net, mask, delta = update_block(net, inp_list, geo_feat, disp, iter16=a.N_GRU_LAYERS == 3, iter08=a.N_GRU_LAYERS >= 2)
            disp = disp + delta.float()         # (same values as the promoting add; fp32 + fp16 takes torch's templated mixed-dtype kernel: 88 us for 14720 elements)
        return {"disp": disp, "mask_feat_4": mask, "net_list": list(net)}
    c = nchw_to_cl
    net = [c(t) for t in net_list]
    inp = [[c(t) for t in ts] for ts in inp_list]
    n_gru = a.N_GRU_LAYERS
    # The disparity lives in three places the loop's kernels read: the NCHW map (lookup), an NHWC [disp, 0, 0, 0] map (the motion encoder's
    # 7x7 convd1) and channel 127 of the 1/4 level's x slot (update.py:92); osa_disp_update_f32 advances all of them with one launch.
    disp = init_disp.float().contiguous().clone()
    disp4 = empty_cl(b, 4, 1, h, w, disp.device)
    if iters < 1:
        raise ValueError("run_refinement: at least one GRU iteration (the reference's loop defines its outputs inside the loop)")
    lvs = update_block._levels(net, inp, n_gru)
    lv4 = lvs[0]
    net = [l.T for l in lvs if l is not None] + net[n_gru:]
    m4t = attach_meta(disp4) if meta_of(lv4.T) is not None else None
    mst = meta_of(lv4.T)
    m4, ms = (None if m4t is None else m4t.data_ptr()), (None if mst is None else mst.data_ptr())
    slot = lv4.T.data_ptr() + 4 * (lv4.hd + 127)
    ext = _ext.load()

    def advance(delta):
        if ext is not None:
            ext.disp_update(disp, delta, 0 if delta is None else delta.shape[1], disp4, lv4.T, lv4.hd + 127, lv4.T.shape[1], b * h * w, m4t, mst)
            return
        _lib.call("osa_disp_update_f32", disp.data_ptr(), None if delta is None else delta.data_ptr(), 0 if delta is None else delta.shape[1],
                  disp4.data_ptr(), slot, lv4.T.shape[1], b * h * w, m4, ms, _stream())
    advance(None)
    mask = None
    for it in range(iters):
        geo_feat = geo_fn.lookup_cl(disp, coords)
        if n_gru == 3 and a.SLOW_FAST_GRU:
            net = update_block.forward_cl(net, inp, iter16=True, iter08=False, iter04=False, update=False)
        if n_gru >= 2 and a.SLOW_FAST_GRU:
            net = update_block.forward_cl(net, inp, iter16=n_gru == 3, iter08=True, iter04=False, update=False)
        net, mask, delta = update_block.forward_cl(net, inp, geo_feat, disp4, iter16=n_gru == 3, iter08=n_gru >= 2,
                                                   want_mask=it == iters - 1, disp_in_place=True)
        advance(delta)
    out = {"disp": disp, "mask_feat_4": cl_to_nchw(mask, 32), "net_list": [cl_to_nchw(t, r.shape[1]) for t, r in zip(net, net_list)]}
    update_block.end_forward()          # the level buffers and the context references die with the forward
    return out
