"""IGEV / StereoBase iterative-refinement update block on the gfx950 engine (BASELINE
configs[4] "IGEV iterative GRU refinement").

Mirror of stereo/modeling/models/igev/update.py:17-144 (stereobase/gru_blocks.py:233-328 is the same
block with a configurable correlation-plane count): same class names, constructor arguments and
state_dict keys; forward on the engine with NHWC tensors:

  * every Conv2d (3x3, 1x1, 7x7, with bias) is the MFMA conv kernel with D = 1; inputs that the
    reference concatenates (`torch.cat([h, x...])`) are channel slices of one buffer,
  * ConvGRU: z = sigmoid(convz(hx) + cz) and q = tanh(convq([r*h, x]) + cq) are fused epilogues
    (residual = cz / cq, OSA_ACT_SIGMOID / OSA_ACT_TANH); r*h = sigmoid(convr(hx) + cr) * h comes out of
    convr's epilogue with h as a raw gate, written straight into the [r*h | x] buffer;
    h' = (1-z)*h + z*q is `osa_gru_combine_f32`,
  * pool2x / interp (avg_pool2d, bilinear align_corners=True) stay PyTorch-ROCm ops on NHWC views
    (tiny maps, feature side).

forward() takes and returns the reference's NCHW tensors; forward_cl() is the channels-last entry.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, amp
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


def pool2x(x):
    """update.py:99-100 on an engine tensor (averages: x's range block stays valid)."""
    return inherit_meta(_as_cl(F.avg_pool2d(x[:, :, 0], 3, stride=2, padding=1)), x)


def interp(x, dest):
    """update.py:107-109 (bilinear, align_corners=True) on engine tensors (convex combinations: x's range block stays valid)."""
    return inherit_meta(_as_cl(F.interpolate(x[:, :, 0], dest.shape[3:], mode="bilinear", align_corners=True)), x)


def _cat_cl(parts, dev, track=False):
    """Channel concatenation into one NHWC buffer (what torch.cat does for the reference).  track (f16x3 chains): the result
    gets a range block = slot-wise maximum of the parts' blocks; a part that has none yet (the static context inputs cz / cr / cq
    ..., converted once per forward) is measured once and keeps its block, so the loop never reduces over data again."""
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

    def forward_cl(self, h, cz, cr, cq, *x_list):
        pz, pr, pq = cached_pack(self, "_eng", lambda: (PackedConv3d(self.convz, None, ACT_SIGMOID),
                                                        PackedConv3d(self.convr, None, ACT_SIGMOID),
                                                        PackedConv3d(self.convq, None, ACT_TANH)))
        hd = self.convz.out_channels
        assert hd % 4 == 0 and h.shape[1] == hd
        f16 = pz.precision == "f16x3"
        hx = _cat_cl([h, *x_list], h.device, track=f16)        # [h | x]
        z = pz(hx, residual=cz)                                # sigmoid(convz(hx) + cz)
        rhx = hx.clone()                                       # [r*h | x]: the x part is shared, r*h overwrites the h slice
        if f16:                                                # range block of [r*h | x]: that of [h | x] itself (|r*h| <= |h|) -- shared,
            rhx._osa_meta = meta_of(hx)                        # not copied: convr folding max |r*h| into it changes nothing
        pr(hx, residual=cr, gate=_nhwc(h), gate_raw=True, out=rhx, out_off=0)   # sigmoid(convr(hx) + cr) * h
        q = pq(rhx, residual=cq)                               # tanh(convq([r*h, x]) + cq)
        out = empty_cl(*h.shape, h.device)
        B, _, _, H, W = h.shape
        _lib.call("osa_gru_combine_f32", z.data_ptr(), q.data_ptr(), h.data_ptr(), out.data_ptr(),
                  B * H * W, hd, z.shape[1], q.shape[1], h.shape[1], out.shape[1], attach_meta(out).data_ptr(), _stream())
        return out

    def forward_train(self, h, cz, cr, cq, *x_list):
        """update.py:36-45: the three 3x3 convolutions on the engine (forward, dgrad, wgrad), gating in torch."""
        with AG.engine_convs():
            x = torch.cat(x_list, dim=1)
            hx = torch.cat([h, x], dim=1)
            z = torch.sigmoid(self.convz(hx) + cz)
            r = torch.sigmoid(self.convr(hx) + cr)
            q = torch.tanh(self.convq(torch.cat([r * h, x], dim=1)) + cq)
        return (1 - z) * h + z * q

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

    def forward_cl(self, disp, corr):
        """disp: engine tensor with the disparity in channel 0 (channels 1..3 zero); corr: engine tensor."""
        R = lambda m: PackedConv3d(m, None, ACT_RELU)
        e = cached_pack(self, "_eng", lambda: dict(c1=R(self.convc1), c2=R(self.convc2), d1=R(self.convd1), d2=R(self.convd2),
                                                   conv=R(self.conv)))
        B, _, _, H, W = disp.shape
        cor_disp = empty_cl(B, 128, 1, H, W, disp.device)      # [cor | disp_]
        e["c2"](e["c1"](corr), out=cor_disp, out_off=0)
        e["d2"](e["d1"](disp), out=cor_disp, out_off=64)
        out = empty_cl(B, 128, 1, H, W, disp.device)           # [conv(cor_disp) (127) | disp]
        e["conv"](cor_disp, out=out, out_off=0)
        out[:, 127] = disp[:, 0]
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
        for m in self.modules():
            if hasattr(m, "_eng"):
                m._eng = None

    def forward_cl(self, net, inp, corr=None, disp=None, iter04=True, iter08=True, iter16=True, update=True):
        """Engine tensors everywhere; `net` (list) is updated in place like the reference does."""
        n_gru = self.args.N_GRU_LAYERS if hasattr(self, "args") else self.n_gru_layers    # igev/update.py vs stereobase/gru_blocks.py
        if iter16:
            net[2] = self.gru16.forward_cl(net[2], *(inp[2]), pool2x(net[1]))
        if iter08:
            if n_gru > 2:
                net[1] = self.gru08.forward_cl(net[1], *(inp[1]), pool2x(net[0]), interp(net[2], net[1]))
            else:
                net[1] = self.gru08.forward_cl(net[1], *(inp[1]), pool2x(net[0]))
        if iter04:
            motion_features = self.encoder.forward_cl(disp, corr)
            if n_gru > 1:
                net[0] = self.gru04.forward_cl(net[0], *(inp[0]), motion_features, interp(net[1], net[0]))
            else:
                net[0] = self.gru04.forward_cl(net[0], *(inp[0]), motion_features)
        if not update:
            return net
        delta_disp = self.disp_head.forward_cl(net[0])
        mask = cached_pack(self, "_mask", lambda: PackedConv3d(self.mask_feat_4[0], None, ACT_RELU), mods=(self.mask_feat_4,))
        return net, mask(net[0]), delta_disp

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
            disp = disp + delta
        return {"disp": disp, "mask_feat_4": mask, "net_list": list(net)}
    c = nchw_to_cl
    net = [c(t) for t in net_list]
    inp = [[c(t) for t in ts] for ts in inp_list]
    disp = init_disp.float()
    mask = None
    for _ in range(iters):
        geo_feat = geo_fn(disp, coords)
        if a.N_GRU_LAYERS == 3 and a.SLOW_FAST_GRU:
            net = update_block.forward_cl(net, inp, iter16=True, iter08=False, iter04=False, update=False)
        if a.N_GRU_LAYERS >= 2 and a.SLOW_FAST_GRU:
            net = update_block.forward_cl(net, inp, iter16=a.N_GRU_LAYERS == 3, iter08=True, iter04=False, update=False)
        geo_feat._osa_meta = geo_fn.meta                  # lookups interpolate / zero-pad the volumes: bounded by their max |.|
        net, mask, delta = update_block.forward_cl(net, inp, c(geo_feat), c(disp),
                                                   iter16=a.N_GRU_LAYERS == 3, iter08=a.N_GRU_LAYERS >= 2)
        disp = disp + cl_to_nchw(delta, 1)
    return {"disp": disp, "mask_feat_4": cl_to_nchw(mask, 32), "net_list": [cl_to_nchw(t, r.shape[1]) for t, r in zip(net, net_list)]}
