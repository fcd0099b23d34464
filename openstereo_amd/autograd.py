"""Differentiable wrappers (training path, SURVEY 8b "Autograd / AMP", Appendix C).

Every Function's forward AND backward run on the engine's HIP kernels:
  * volume constructors  -> osa_build_volume_f32 / osa_build_volume_bwd_f32
  * regression heads     -> osa_*softargmin*_f32 / *_bwd_f32
  * Conv3d / ConvTranspose3d / Conv2d -> the MFMA implicit-GEMM kernel (forward), the SAME kernel with
    role-swapped / flipped weight packing for the data gradient, and the fp32-MFMA wgrad kernel for the
    weight gradient.
BatchNorm (batch statistics, SyncBN), activations and residual adds stay ordinary torch modules in
training mode, so their semantics (running-stat updates, DDP/SyncBN hooks) are exactly the reference's;
fusion of BN/activation into the conv epilogue is an inference-only optimisation.
Gradients flow to nn.Parameters as usual, so DistributedDataParallel's bucketed RCCL all-reduce works
unchanged (one process per GPU).
"""
from __future__ import annotations

import ctypes
import math
import os
import threading

import torch

from . import _lib, amp, engine, ops, timing
from .ops import _f32c, _p, _stream, channel_sums, cl_rows, empty_cl, is_cl, to_cl
from .ranges import input_meta, attach_meta


# Every Function runs its forward with autocast switched off and floating inputs cast to fp32 (torch.amp.custom_fwd): the kernels compute
# in fp32-class arithmetic whatever the surrounding region says; the callers below cast the result to the dtype the reference's op would
# have produced there (openstereo_amd/amp.py).  Backward runs under the same (disabled) autocast state (custom_bwd).
_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = torch.amp.custom_bwd(device_type="cuda")


# ----------------------------------------------------------------------------- volumes
class _GwcVolume(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, left, right, maxdisp, num_groups):
        l, r = _f32c(left), _f32c(right)
        ctx.save_for_backward(l, r)
        ctx.meta = (maxdisp, num_groups, left.dtype)
        return ops._build(l, r, num_groups, None, None, maxdisp, ops.NCDHW)

    @staticmethod
    @_bwd
    def backward(ctx, dvol):
        l, r = ctx.saved_tensors
        maxdisp, G, dt = ctx.meta
        B, C, H, W = l.shape
        dv = _f32c(dvol)
        ext = engine._ext.load()
        if ext is not None:
            dl, dr = ext.volume_bwd(dv, l, r, [B, C, H, W], maxdisp, G, False, True)
            return dl.to(dt), dr.to(dt), None, None
        dl, dr = torch.empty_like(l), torch.empty_like(r)
        _lib.call("osa_build_volume_bwd_f32", dv.data_ptr(), l.data_ptr(), r.data_ptr(), dl.data_ptr(), dr.data_ptr(),
                  B, C, H, W, maxdisp, G, 0, 1, G, 0, _stream())
        return dl.to(dt), dr.to(dt), None, None


class _ConcatVolume(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, left, right, maxdisp, mask_left):
        l, r = _f32c(left), _f32c(right)
        ctx.meta = (maxdisp, mask_left, left.dtype, tuple(l.shape))
        return ops._build(None, None, 0, l, r, maxdisp, ops.NCDHW, mask_left=mask_left)

    @staticmethod
    @_bwd
    def backward(ctx, dvol):
        maxdisp, mask_left, dt, (B, C, H, W) = ctx.meta
        dv = _f32c(dvol)
        ext = engine._ext.load()
        if ext is not None:
            dl, dr = ext.volume_bwd(dv, None, None, [B, C, H, W], maxdisp, 0, True, bool(mask_left))
            return dl.to(dt), dr.to(dt), None, None
        dl = torch.empty((B, C, H, W), device=dv.device, dtype=torch.float32)
        dr = torch.empty_like(dl)
        _lib.call("osa_build_volume_bwd_f32", dv.data_ptr(), None, None, dl.data_ptr(), dr.data_ptr(),
                  B, C, H, W, maxdisp, 0, 1, 1 if mask_left else 0, 2 * C, 0, _stream())
        return dl.to(dt), dr.to(dt), None, None


def build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    assert refimg_fea.shape[1] % num_groups == 0
    return _GwcVolume.apply(refimg_fea, targetimg_fea, maxdisp, num_groups)


def build_concat_volume(refimg_fea, targetimg_fea, maxdisp, mask_left=True):
    return _ConcatVolume.apply(refimg_fea, targetimg_fea, maxdisp, mask_left)


def correlation_volume(left_feature, right_feature, max_disp):
    return _GwcVolume.apply(left_feature, right_feature, max_disp, 1)[:, 0]


# ----------------------------------------------------------------------------- regression
class _SoftArgmin(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, prob):
        ctx.shape = tuple(prob.shape)
        return ops.disparity_regression(prob, prob.shape[1], keepdim=False)

    @staticmethod
    @_bwd
    def backward(ctx, dout):
        B, D, H, W = ctx.shape
        g = _f32c(dout)
        ext = engine._ext.load()
        if ext is not None:
            return ext.softargmin_bwd(g, D)
        dp = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        _lib.call("osa_softargmin_bwd_f32", g.data_ptr(), dp.data_ptr(), B, D, H, W, _stream())
        return dp


class _SoftmaxSoftArgmin(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, cost):
        c = _f32c(cost)
        ctx.save_for_backward(c)
        return ops.softmax_disparity_regression(c, keepdim=False)

    @staticmethod
    @_bwd
    def backward(ctx, dout):
        (c,) = ctx.saved_tensors
        B, D, H, W = c.shape
        g = _f32c(dout)
        ext = engine._ext.load()
        if ext is not None:
            return ext.softmax_softargmin_bwd(c, g)
        dc = torch.empty_like(c)
        _lib.call("osa_softmax_softargmin_bwd_f32", c.data_ptr(), g.data_ptr(), dc.data_ptr(), B, D, H, W, _stream())
        return dc


class _UpsampleSoftArgmin(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, cost_low, maxdisp, h, w, align_corners):
        c = _f32c(cost_low)
        ctx.save_for_backward(c)
        ctx.meta = (maxdisp, h, w, align_corners)
        return ops.upsample_softargmin(c, maxdisp, h, w, align_corners)

    @staticmethod
    @_bwd
    def backward(ctx, dout):
        (c,) = ctx.saved_tensors
        maxdisp, h, w, align = ctx.meta
        B, Dl, Hl, Wl = c.shape
        g = _f32c(dout)
        ext = engine._ext.load()
        if ext is not None:
            return ext.upsample_softargmin_bwd(c, g, int(maxdisp), int(h), int(w), bool(align)), None, None, None, None
        dc = torch.empty_like(c)
        need = _lib.load().osa_upsample_softargmin_bwd_workspace_bytes(B, Dl, int(h), int(w))     # [B,Dl,H,W] scratch: atomic-free, deterministic
        ws = torch.empty((need + 3) // 4, device=c.device, dtype=torch.float32)
        _lib.call("osa_upsample_softargmin_bwd_ws_f32", c.data_ptr(), g.data_ptr(), dc.data_ptr(), B, Dl, Hl, Wl,
                  int(maxdisp), int(h), int(w), 1 if align else 0, ws.data_ptr(), need, _stream())
        return dc, None, None, None, None


def disparity_regression(prob, maxdisp, keepdim=True):
    assert len(prob.shape) == 4 and prob.shape[1] == maxdisp
    out = _SoftArgmin.apply(prob)
    return out.unsqueeze(1) if keepdim else out


def softmax_disparity_regression(cost, keepdim=True):
    out = _SoftmaxSoftArgmin.apply(cost)
    return out.unsqueeze(1) if keepdim else out


def upsample_softargmin(cost_lowres, maxdisp, h, w, align_corners=False):
    if cost_lowres.dim() == 5:
        cost_lowres = cost_lowres[:, 0]
    return _UpsampleSoftArgmin.apply(cost_lowres, maxdisp, h, w, align_corners)


# ----------------------------------------------------------------------------- convolutions
def _wcache(w):
    """Per-weight pack cache handle `(dict, stamp)`.  The dict lives ON the weight tensor object (the nn.Parameter the caller holds), so
    it dies with the model -- a process-wide table keyed on data_ptr would serve a freed model's packs to the next one the allocator
    places there.  The stamp identifies the CONTENT the packs were built from and is read from the caller's tensor itself --
    (_version, data_ptr, device, dtype) of the Parameter, not of the fp32-contiguous working copy `_f32c` may have to make (a fresh
    copy always has version 0: with channels_last / fp16 / bf16 parameters the memo would never invalidate after an optimizer step;
    ADVICE r2).  `.to(other device)`, `.double()`, `load_state_dict` and every in-place update through autograd-visible ops change the
    stamp.  Not covered: writes through `.data` / raw pointers, which leave `_version` untouched -- call `engine.reset_engine(model)`
    or `weight._osa_packs.clear()` after such an update."""
    c = getattr(w, "_osa_packs", None)
    if c is None:
        try:
            w._osa_packs = c = {}
        except Exception:                               # tensor subclasses without a __dict__
            return None
    return _Memo(c, (w._version, w.data_ptr(), str(w.device), w.dtype))


class _Memo:
    """(dict, stamp) handle.  A plain object on purpose: torch.amp.custom_fwd walks tuples / dicts among a Function's arguments and
    rebuilds them (casting tensors inside), which would hand the Function a COPY of the memo dict."""
    __slots__ = ("memo", "stamp")

    def __init__(self, memo, stamp):
        self.memo, self.stamp = memo, stamp


def _pack(w, Ci, Co, k, mode, precision, cache=None):
    """Packed image of a weight for one role, memoised per weight object and content stamp: a layer that runs many times between two
    optimizer steps -- the update block: 22 GRU iterations, forward and data gradient each -- is packed (and its power-of-two scale
    measured, a host sync) once per role and step instead of once per call.  `w` is the fp32-contiguous working tensor, `cache` the
    `_wcache` handle of the tensor it came from."""
    if cache is None:
        return _pack_now(w, Ci, Co, k, mode, precision)
    memo, stamp = cache.memo, cache.stamp
    if memo.get("stamp") != stamp:
        memo.clear()
        memo["stamp"] = stamp
    key = (tuple(w.shape), mode, precision)
    hit = memo.get(key)
    if hit is None:
        hit = memo[key] = _pack_now(w, Ci, Co, k, mode, precision)
    return hit


def _memo(cache, key, build):
    """`build()` memoised in a weight's pack cache under the same content stamp as its packs (see _pack)"""
    if cache is None:
        return build()
    memo, stamp = cache.memo, cache.stamp
    if memo.get("stamp") != stamp:
        memo.clear()
        memo["stamp"] = stamp
    hit = memo.get(key)
    if hit is None:
        hit = memo[key] = build()
    return hit


def _wcache2(w, w2):
    """pack-cache handle of a PAIR of weights packed as one layer: the dict lives on the first, the stamp covers both"""
    a, b = _wcache(w), _wcache(w2)
    if a is None or b is None:
        return None
    pair = a.memo.get(("pair", id(w2)))                 # a dict of its own: the pair's stamp must not overwrite the stamp of w used alone
    if pair is None:
        pair = a.memo[("pair", id(w2))] = {}
    return _Memo(pair, (a.stamp, b.stamp))


def _pack_now(w, Ci, Co, k, mode, precision):
    """mode: 'fwd' | 'dgrad_s1' (role swap + flip) | 'dgrad_of_deconv' (role swap) | 'deconv' / 'deconv2d' (parity-class packing).
    Returns (packed buffer, scale): f32 -> scale 1.0; f16x3 -> a 2-float DEVICE tensor {wscale, 1 / wscale} that the pack kernel derived
    from max |w| on the device (osa_*_pack_*_auto) -- no host synchronisation per layer / role / optimizer step, and the whole training
    step can be captured in a hipGraph (the packs are part of the captured work)."""
    f16 = precision == "f16x3"
    h16 = precision == "f16"            # native fp16 operands (AMP training, r5): weights rounded to nearest even, no scale
    lib = _lib.load()
    ext = engine._ext.load()
    if ext is not None:
        # PyTorch-ROCm C++ extension (csrc/torch_ext.cpp conv_pack / deconv_pack): the same pack kernels, one dispatcher call
        amax = sc = None
        if f16:
            amax = torch.linalg.vector_norm(w.detach(), float("inf")).reshape(1)
            sc = torch.empty(2, device=w.device, dtype=torch.float32)
        pid = _PREC_ID[precision]
        if mode in ("deconv2d", "deconv"):
            n = (lib.osa_deconv2d_packed_floats if mode == "deconv2d" else lib.osa_deconv3d_packed_floats)(Ci, Co, k[0])
            buf = torch.zeros(n, device=w.device, dtype=torch.float32)
            ext.deconv_pack(w, buf, [Ci, Co, k[0], 1, 1 if mode == "deconv2d" else 0], pid, amax, sc)
        else:
            n = lib.osa_conv3d_packed_floats(Ci, Co, *k)
            buf = torch.zeros(n, device=w.device, dtype=torch.float32)
            tr, fl = {"fwd": (0, 0), "dgrad_s1": (1, 1), "dgrad_of_deconv": (0, 0)}[mode]
            ext.conv_pack(w, buf, [Ci, Co, k[0], k[1], k[2], tr, fl], pid, amax, sc)
        return buf, (sc if f16 else 1.0)
    if f16:
        amax = torch.linalg.vector_norm(w.detach(), float("inf"))          # one reduction kernel, stays on the device
        sc = torch.empty(2, device=w.device, dtype=torch.float32)
    if mode == "deconv2d":
        n = lib.osa_deconv2d_packed_floats(Ci, Co, k[0])
        buf = torch.zeros(n, device=w.device, dtype=torch.float32)
        if f16:
            _lib.call("osa_deconv2d_pack_f16x3_auto", w.data_ptr(), buf.data_ptr(), Ci, Co, k[0], 1, amax.data_ptr(), sc.data_ptr(), _stream())
        elif h16:
            _lib.call("osa_deconv2d_pack_f16", w.data_ptr(), buf.data_ptr(), Ci, Co, k[0], 1, _stream())
        else:
            _lib.call("osa_deconv2d_pack_f32", w.data_ptr(), buf.data_ptr(), Ci, Co, k[0], 1, _stream())
    elif mode == "deconv":
        n = lib.osa_deconv3d_packed_floats(Ci, Co, k[0])
        buf = torch.zeros(n, device=w.device, dtype=torch.float32)
        if f16:
            _lib.call("osa_deconv3d_pack_f16x3_auto", w.data_ptr(), buf.data_ptr(), Ci, Co, k[0], 1, amax.data_ptr(), sc.data_ptr(), _stream())
        elif h16:
            _lib.call("osa_deconv3d_pack_f16", w.data_ptr(), buf.data_ptr(), Ci, Co, k[0], 1, _stream())
        else:
            _lib.call("osa_deconv3d_pack_f32", w.data_ptr(), buf.data_ptr(), Ci, Co, k[0], 1, _stream())
    else:
        n = lib.osa_conv3d_packed_floats(Ci, Co, *k)
        buf = torch.zeros(n, device=w.device, dtype=torch.float32)
        tr, fl = {"fwd": (0, 0), "dgrad_s1": (1, 1), "dgrad_of_deconv": (0, 0)}[mode]
        if f16:
            _lib.call("osa_conv3d_pack_ex_auto", w.data_ptr(), buf.data_ptr(), Ci, Co, *k, tr, fl, amax.data_ptr(), sc.data_ptr(), _stream())
        else:
            _lib.call("osa_conv3d_pack_ex", w.data_ptr(), buf.data_ptr(), Ci, Co, *k, tr, fl, 2 if h16 else 0, 1.0, _stream())
    return buf, (sc if f16 else 1.0)


def _ranges(x, y, wscale, xmeta=None):
    """f16x3 operand ranges of a plain conv call: x (activations or incoming gradients -- whose magnitudes are
    routinely 1e-5 .. 1e-8) is scaled by a power of two derived from its measured max |.| on the device, so the
    hi/lo fp16 halves keep 22 significant bits whatever the magnitude; exact to undo.  wscale: the packed weights' device-side
    {scale, 1 / scale} pair (_pack_now)."""
    return _lib.F16x3Ranges((xmeta if xmeta is not None else input_meta(x)).data_ptr(), None, None, attach_meta(y).data_ptr(), None, None, wscale.data_ptr())


_PREC_ID = {"f32": 0, "f16x3": 1, "f16": 2}


def _ext_conv(family, x, packed, y, dims, geom, precision, oscale, xmeta=None, scale=None, shift=None):
    """The plain conv / transposed-conv launch of the training path through the PyTorch-ROCm C++ extension (torch.ops.osa_native.conv_ndhwc:
    one dispatcher call, tensors in, current HIP stream inside) -- False when the extension is not loaded (the caller then goes through
    ctypes; same entry point of the C ABI, bit-identical)."""
    ext = engine._ext.load()
    if ext is None:
        return False
    metas = []
    if precision == "f16x3":
        e = engine._empty(x.device)
        metas = [xmeta if xmeta is not None else input_meta(x), e, e, attach_meta(y), e, e, oscale]
    ext.conv_ndhwc(x, 0, packed, scale, shift, None, 0, y, 0, None, dims, geom, family, _PREC_ID[precision], 0, 0.0, 1.0, metas)
    return True


def _sfx_tail(precision, x, y, oscale, xmeta=None):
    """entry-point suffix and trailing arguments of a plain conv / deconv call in the given arithmetic mode"""
    if precision == "f16x3":
        return "f16x3", (1.0, _ranges(x, y, oscale, xmeta), _stream())
    return ("f16" if precision == "f16" else "f32"), (_stream(),)        # f16: fp32 NDHWC tensors, fp16 operands (no flags, no ranges)


def _alias(t):
    """a tensor object over t's storage and geometry that autograd does not regard as a view of anything"""
    return torch.empty(0, device=t.device, dtype=t.dtype).set_(t.untyped_storage(), t.storage_offset(), t.shape, t.stride())


_ONES = {}


def _ones(n, device):
    """cached all-ones scale vector: `y = acc * 1 + bias` puts a convolution's bias into the launch's epilogue (scale and shift travel together)"""
    t = _ONES.get((n, device))
    if t is None:
        t = _ONES[(n, device)] = torch.ones(n, device=device, dtype=torch.float32)
    return t


def _run_conv(x, packed, oscale, Ci, Co, k, stride, pad, dil, precision, out_shape, xmeta=None, bias=None):
    """x NDHWC (channels padded to 4). plain conv; bias (fp32 [Co]) is added in the epilogue."""
    B, Cs, D, H, W = x.shape
    CoS = (Co + 3) // 4 * 4
    y = empty_cl(B, CoS, *out_shape, x.device)
    if CoS != Co:
        y.zero_()
    Ci4 = (Ci + 3) // 4 * 4
    macs = B * out_shape[0] * out_shape[1] * out_shape[2] * Ci * Co * k[0] * k[1] * k[2]
    with timing.span("conv3d", Ci, Co, k[1], stride, D, H, W, flops=2 * macs,
                     nbytes=4 * B * (D * H * W * Ci + out_shape[0] * out_shape[1] * out_shape[2] * Co)):
        scale = None if bias is None else _ones(Co, x.device)
        if not _ext_conv(0, x, packed, y, [B, D, H, W, Ci4, Cs, Co, CoS, 0, 0],
                         [k[0], k[1], k[2], stride, pad[0], pad[1], pad[2], dil[0], dil[1], dil[2]], precision, oscale, xmeta, scale, bias):
            sfx, tail = _sfx_tail(precision, x, y, oscale, xmeta)
            _lib.call("osa_conv3d_ndhwc_" + sfx, x.data_ptr(), packed.data_ptr(), _p(scale), _p(bias), None, y.data_ptr(),
                      B, D, H, W, Ci4, Cs, Co, CoS, 0, k[0], k[1], k[2], stride, pad[0], pad[1], pad[2], dil[0], dil[1], dil[2],
                      None, 0, 0, 0.0, *tail)
    return y


def _run_deconv(x, packed, oscale, Ci, Co, k, pad, opad, precision):
    B, Cs, D, H, W = x.shape
    CoS = (Co + 3) // 4 * 4
    od = lambda n: (n - 1) * 2 - 2 * pad + k + opad
    y = empty_cl(B, CoS, od(D), od(H), od(W), x.device)
    if CoS != Co:
        y.zero_()
    Ci4 = (Ci + 3) // 4 * 4
    with timing.span("deconv3d", Ci, Co, k, 2, D, H, W, flops=2 * B * D * H * W * Ci * Co * k ** 3,
                     nbytes=4 * B * (D * H * W * Ci + od(D) * od(H) * od(W) * Co)):
        if not _ext_conv(1, x, packed, y, [B, D, H, W, Ci4, Cs, Co, CoS, 0, 0], [k, pad, opad], precision, oscale):
            sfx, tail = _sfx_tail(precision, x, y, oscale)
            _lib.call("osa_deconv3d_ndhwc_" + sfx, x.data_ptr(), packed.data_ptr(), None, None, None, y.data_ptr(),
                      B, D, H, W, Ci4, Cs, Co, CoS, 0, k, pad, opad, None, 0, 0, 0.0, *tail)
    return y


def _out(n, k, p, d, s):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


WGRAD_F16X3 = True      # f16x3 mode: weight gradients of the layers osa_conv3d_wgrad_ws_f16x3 covers on the split-precision kernel (False: always exact fp32)
# The same kernel also covers stride-2 / transposed layers (class mode: parity sub-lattices, one tap group per d delta).  Correct (tests) but not
# faster than the fp32 class-mode kernel, which takes a whole class of up to 8 taps per staged brick where this form needs two groups of
# <= 4 (measured: GwcNet step 33.9 -> 34.4 ms, StereoBase whole model 176 -> 184 ms): off by default.
WGRAD_F16X3_CLASS = False
# native f16 mode (AMP training, r5): the same kernel with fp16 operands and one MFMA per product.  Class mode (stride-2 / transposed layers)
# is on there: with a third of the matrix work and half the LDS footprint it no longer trails the fp32 class-mode kernel (to be re-measured).
WGRAD_F16 = True
WGRAD_F16_CLASS = os.environ.get("OSA_WGRAD_F16_CLASS", "1") != "0"


def _wgrad(xc, dyc, dw, B, D, H, W, Ci, Do, Ho, Wo, Co, k, stride, pad, dil, transposed, precision="f32", xmeta=None, dymeta=None, xcs=None, dycs=None):
    """Weight gradient, two-stage form: partial tiles in a scratch tensor from the caching allocator, summed in a fixed order --
    deterministic, and free of the contended float atomics of the one-stage form.  precision "f16x3": the split-precision kernel where it
    applies (unit stride / dilation, 3x3 planes: osa_conv3d_wgrad_ws_f16x3, operand ranges from the tensors' range blocks), the exact
    fp32 kernel (osa_conv3d_wgrad_ws_f32) everywhere else.  fp16 tensors (the native f16 form) that the fp16 kernel declines are widened
    and take the exact kernel (never raise inside backward() for a layer the forward accepted).
    xc / dyc may be LISTS of equally shaped and strided tensors (the queued uses of one weight, r6): one launch over all of them without a
    concatenation (osa_conv3d_wgrad_ws_multi); B is then the total batch and f16x3 range blocks must cover all items."""
    multi = isinstance(xc, (list, tuple))
    xs, dys = (list(xc), list(dyc)) if multi else ([xc], [dyc])
    xc, dyc = xs[0], dys[0]
    assert precision == "f16" or (xc.dtype == torch.float32 and dyc.dtype == torch.float32), "fp16 tensors exist in the native f16 form only"
    xcs, dycs = (xc.shape[1] if xcs is None else xcs), (dyc.shape[1] if dycs is None else dycs)        # channel strides (rows may be wider than the tensors' logical channels)
    dims = (B, D, H, W, Ci, Do, Ho, Wo, Co, k[0], k[1], k[2], stride, pad[0], pad[1], pad[2], dil[0], dil[1], dil[2], transposed)
    lib = _lib.load()
    vox = (D * H * W) if transposed else (Do * Ho * Wo)            # positions every weight tap is accumulated over
    span = dict(flops=2 * B * vox * Ci * Co * k[0] * k[1] * k[2], nbytes=4 * B * (D * H * W * Ci + Do * Ho * Wo * Co))
    if precision == "f16x3" and (xmeta is None or dymeta is None):
        from .ranges import combine_meta
        xmeta = xmeta if xmeta is not None else combine_meta(*[input_meta(t) for t in xs])
        dymeta = dymeta if dymeta is not None else combine_meta(*[input_meta(t) for t in dys])
    ext = engine._ext.load()
    if ext is not None:
        # PyTorch-ROCm C++ extension (csrc/torch_ext.cpp conv_wgrad): workspace query, allocation and launch in one dispatcher call
        ed = [B, D, H, W, Ci, xcs, Do, Ho, Wo, Co, dycs, k[0], k[1], k[2], stride, pad[0], pad[1], pad[2], dil[0], dil[1], dil[2], transposed]
        run = (lambda prec, mx, md: ext.conv_wgrad_multi(xs, dys, dw, ed, prec, mx, md)) if multi else (lambda prec, mx, md: ext.conv_wgrad(xs[0], dys[0], dw, ed, prec, mx, md))
        if precision == "f16" and WGRAD_F16 and ((stride == 1 and not transposed) or WGRAD_F16_CLASS):
            with timing.span("wgrad_f16", Ci, Co, k[1], stride, D, H, W, transposed, **span):
                if run(2, None, None):
                    return
        if precision == "f16x3" and WGRAD_F16X3 and ((stride == 1 and not transposed) or WGRAD_F16X3_CLASS):
            with timing.span("wgrad_f16x3", Ci, Co, k[1], stride, D, H, W, transposed, **span):
                if run(1, xmeta, dymeta):
                    return
        if xc.dtype != torch.float32 or dyc.dtype != torch.float32:
            xs, dys = [t.float() for t in xs], [t.float() for t in dys]      # (dense tensors: .float() keeps the strides the channel strides refer to)
        with timing.span("wgrad", Ci, Co, k[1], stride, D, H, W, transposed, **span):
            run(0, None, None)
        return

    def launch(form, mx, md, need):
        ws = torch.empty((need + 3) // 4, device=xc.device, dtype=torch.float32)
        ptrs = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        _lib.call("osa_conv3d_wgrad_ws_multi", form, ptrs(xs), ptrs(dys), len(xs), dw.data_ptr(), B, D, H, W, Ci, xcs,
                  Do, Ho, Wo, Co, dycs, k[0], k[1], k[2], stride, pad[0], pad[1], pad[2], dil[0], dil[1], dil[2], transposed,
                  _p(mx), _p(md), int(xs[0].dtype == torch.float16), int(dys[0].dtype == torch.float16), ws.data_ptr(), need, _stream())
    if precision == "f16" and WGRAD_F16 and ((stride == 1 and not transposed) or WGRAD_F16_CLASS):
        # native fp16 operands, one MFMA per product, no range blocks (AMP training: GradScaler owns the range)
        need = lib.osa_conv3d_wgrad_f16x3_workspace_bytes(*dims)
        if need:
            with timing.span("wgrad_f16", Ci, Co, k[1], stride, D, H, W, transposed, **span):
                launch(2, None, None, need)
            return
    if precision == "f16x3" and WGRAD_F16X3 and ((stride == 1 and not transposed) or WGRAD_F16X3_CLASS):
        need = lib.osa_conv3d_wgrad_f16x3_workspace_bytes(*dims)
        if need:
            with timing.span("wgrad_f16x3", Ci, Co, k[1], stride, D, H, W, transposed, **span):
                launch(1, xmeta, dymeta, need)
            return
    need = lib.osa_conv3d_wgrad_workspace_bytes(*dims)
    if need == 0:
        raise _lib.EngineError("osa_conv3d_wgrad_workspace_bytes: unsupported layer " + str(dims))
    if xc.dtype != torch.float32 or dyc.dtype != torch.float32:
        xs, dys = [t.float() for t in xs], [t.float() for t in dys]
    with timing.span("wgrad", Ci, Co, k[1], stride, D, H, W, transposed, **span):
        launch(0, None, None, need)


# ----------------------------------------------------------------------------- deferred, batched weight gradients (r6)
# A weight that is applied several times in one step -- the update block's convolutions: 22 GRU iterations -- used to get 22 weight-gradient
# launches of one small map each (14720 pixels at 320x736: ~100 workgroups, most of the launch is staging latency) plus 21 accumulation adds.
# Now every differentiable use is counted in forward; the backward calls queue their (x, dy) pairs and return no weight gradient until the
# LAST pending use arrives, which runs ONE weight-gradient launch over the batch of all queued pairs and returns the total -- the same
# sum, delivered through autograd as before (one AccumulateGrad, so DistributedDataParallel sees the parameter ready once).  The state lives
# in the weight's pack memo (it dies with the content stamp: an optimizer step starts from zero).  Safety net: a use whose backward never
# runs (a differentiable forward that is not back-propagated) would leave the count short; an engine callback at the end of every backward
# pass therefore delivers whatever is still queued straight into `.grad`.
DEFER_WGRAD = os.environ.get("OSA_DEFER_WGRAD", "1") != "0"
# training-mode BatchNorm (batch statistics) on channels-last tensors as osa_channel_sums + osa_channel_affine passes (_TrainBN): correct (tests),
# but OFF by default -- the per-channel coefficient arithmetic between the passes is ~20 tiny torch launches per layer and step, and the
# GwcNet training step measured 39.9 ms with it against 35.8 ms with torch's own kernels (profiles/round6/training_steps_r6.txt).
TRAIN_BN = os.environ.get("OSA_TRAIN_BN", "0") == "1"
FROZEN_BN = os.environ.get("OSA_FROZEN_BN", "1") != "0"      # eval-mode BatchNorm modules inside engine_convs: backward as one osa_channel_sums pass
_defer_live = []           # states with queued pairs in the running backward pass
_defer_cb = threading.local()


def _defer_note_use(cache, need, w, w2, flat, co1, bias=None):
    if cache is None or not DEFER_WGRAD or not need:
        return
    st = cache.memo.get("defer")
    if st is None:
        import weakref
        st = cache.memo["defer"] = {"uses": 0, "done": 0, "pend": {}, "w": weakref.ref(w), "w2": None if w2 is None else weakref.ref(w2), "flat": flat, "co1": co1,
                                    "b": None if bias is None else weakref.ref(bias)}
    st["uses"] += 1


def _defer_run(pend):
    """pend: {key: (run, [items])} -- one launch per group of equally shaped uses (a weight applied at two resolutions, or through two
    Functions, has one group per form); the groups' results (dw, db) are summed in insertion order."""
    dw = db = None
    for run, items in pend.values():
        g, gb = run(items)
        dw = g if dw is None or g is None else dw + g
        db = gb if db is None or gb is None else db + gb
    return dw, db


def _defer_final_flush():
    _defer_cb.armed = False
    for st in list(_defer_live):
        pend = st["pend"]
        st.update(uses=0, done=0, pend={})
        if not pend:
            continue
        dw, db = _defer_run(pend)                              # uses of this weight that were not part of the finished backward pass stay uncounted
        parts = []
        if dw is not None:
            if st["flat"]:
                dw = dw[:, :, 0]
            parts.append((st["w"](), dw if st["co1"] is None else dw[:st["co1"]]))
            if st["w2"] is not None:
                parts.append((st["w2"](), dw[st["co1"]:]))
        if db is not None and st["b"] is not None:
            parts.append((st["b"](), db))
        for prm, g in parts:
            if prm is not None and prm.requires_grad and prm.is_leaf:
                g = g.to(prm.dtype)
                prm.grad = g.clone() if prm.grad is None else prm.grad + g
    _defer_live.clear()


def _defer_wgrad(cache, key, item, run):
    """item: one (x, dy, ...) pair of a backward call; key: everything run() takes from its closure (shapes, arithmetic mode, strides);
    run(items) -> (dw, db) over all items of one key (either may be None).  Returns the gradients of every queued use when this is the
    last one the forward counted, (None, None) before that (autograd reads None as zero)."""
    st = None if cache is None else cache.memo.get("defer")
    if st is None or st["uses"] <= 1:
        if st is not None:
            st.update(uses=0, done=0, pend={})
        return run([item])
    st["pend"].setdefault(key, (run, []))[1].append(item)
    st["done"] += 1
    if st["done"] < st["uses"]:
        if st not in _defer_live:
            _defer_live.append(st)
        if not getattr(_defer_cb, "armed", False):
            _defer_cb.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(_defer_final_flush)
        return None, None
    pend = st["pend"]
    st.update(uses=0, done=0, pend={})
    if st in _defer_live:
        _defer_live.remove(st)
    return _defer_run(pend)


MULTI_WGRAD = os.environ.get("OSA_WGRAD_MULTI", "1") != "0"   # queued uses of one weight: ONE launch over the list of tensors (0: concatenate them first)


def _same_layout(ts):
    t0 = ts[0]
    return all(t.shape == t0.shape and t.stride() == t0.stride() and t.dtype == t0.dtype and t.device == t0.device for t in ts[1:])


def _bias_grad_items(dys, Co):
    """bias gradient over a list of queued gradients: per-item sums added up (22 launches of ~8 us, no concatenation)"""
    gs = [t[:, :Co] for t in dys]
    if 1 < len(gs) <= 24 and _same_layout(gs) and cl_rows(gs[0]) is not None:
        return ops.channel_sums_list(gs)
    tot = None
    for t in dys:
        g = _bias_grad(t, Co)
        tot = g if tot is None else tot + g
    return tot


def _bias_grad(dyc, Co):
    """sum of an NDHWC gradient over its positions, per channel (fp32): one coalesced pass of the engine's kernel where the layout allows"""
    g = dyc[:, :Co]
    if cl_rows(g) is not None:
        return channel_sums(g)[0][0]
    return g.sum((0, 2, 3, 4), dtype=torch.float32)


def _cat_batch(ts, C=None):
    """the batch concatenation of equally shaped NDHWC tensors (logical channels [0, C) only when their rows are wider): one copy"""
    if len(ts) == 1 and C is None:
        return ts[0]
    return torch.cat([t.permute(0, 2, 3, 4, 1) if C is None else t.permute(0, 2, 3, 4, 1)[..., :C] for t in ts], 0).permute(0, 4, 1, 2, 3)

class _Conv3d(torch.autograd.Function):
    """y = conv3d(x, w) (no bias).  x: logical [B,Ci,D,H,W] (any strides); y: NDHWC-strided [B,Co,...]."""

    @staticmethod
    @_fwd
    def forward(ctx, x, w, stride, pad, dil, precision, cache=None, w2=None, bias=None, flat=False):
        # flat: the 2-D case -- x is [B,Ci,H,W], w [Co,Ci,kh,kw] (the D = 1 case of the same kernels), the result [B,Co,H,W]
        ctx.flat = flat
        if flat:
            x, w = x.unsqueeze(2), w.unsqueeze(2)
            w2 = None if w2 is None else w2.unsqueeze(2)
        xc = to_cl(x)                               # NDHWC, channels padded to a multiple of 4 with zeros
        # w2: a second layer over the same input, stacked on the output axis -- ONE forward / data-gradient / weight-gradient launch for
        # both (ConvGRU's convz | convr read the same [h | x], update.py:38-39).  The concatenation is memoised with the packs.
        ctx.co1 = None if w2 is None else w.shape[0]
        wf = _f32c(w) if w2 is None else _memo(cache, "wcat", lambda: torch.cat([_f32c(w.detach()), _f32c(w2.detach())], 0))
        Co, Ci = wf.shape[:2]
        k = tuple(wf.shape[2:])
        packed, osc = _pack(wf, Ci, Co, k, "fwd", precision, cache)
        B, _, D, H, W = xc.shape
        sd = 1 if (D == 1 and k[0] == 1) else stride
        oshape = (_out(D, k[0], pad[0], dil[0], sd), _out(H, k[1], pad[1], dil[1], stride), _out(W, k[2], pad[2], dil[2], stride))
        # one max |x| reduction serves the forward conv and (ctx.xmeta) the f16x3 weight gradient; kept on ctx, not on the tensor object
        xmeta = input_meta(xc) if precision == "f16x3" else None
        # bias: added in the conv launch's epilogue (r5: one elementwise launch less per biased layer and call -- ~200 per StereoBase step)
        ctx.bias_dt = None if bias is None else bias.dtype
        y = _run_conv(xc, packed, osc, Ci, Co, k, stride, pad, dil, precision, oshape, xmeta, None if bias is None else _f32c(bias.detach()))
        ctx.save_for_backward(xc, wf)
        ctx.xmeta = xmeta
        ctx.meta = (stride, pad, dil, precision, tuple(x.shape), x.dtype)
        ctx.cache = cache
        _defer_note_use(cache, ctx.needs_input_grad[1] or (w2 is not None and ctx.needs_input_grad[7]) or (bias is not None and ctx.needs_input_grad[8]),
                        w, w2, flat, ctx.co1, bias)
        # The result is a channel slice (and, flat, a squeeze) of the internal NDHWC allocation.  Returned as a view, autograd refuses
        # in-place operations on it ("view created inside a custom Function") -- and the reference's modules put nn.ReLU(inplace=True)
        # right behind biased convolutions (update.py:19-26); the bias add used to give them a fresh tensor.  So: a tensor object over the
        # same storage and geometry that is not a view.
        return _alias(y[:, :Co, 0] if flat else y[:, :Co])

    @staticmethod
    @_bwd
    def backward(ctx, dy):
        xc, wf = ctx.saved_tensors
        stride, pad, dil, precision, xshape, xdt = ctx.meta
        Co, Ci = wf.shape[:2]
        k = tuple(wf.shape[2:])
        if ctx.flat:
            dy = dy.unsqueeze(2)
        dyc = to_cl(dy)
        dymeta = input_meta(dyc) if precision == "f16x3" else None       # shared by the data gradient and the weight gradient
        B, _, D, H, W = xc.shape
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if stride == 1:
                packed, osc = _pack(wf, Co, Ci, k, "dgrad_s1", precision, ctx.cache)
                p2 = tuple(dil[i] * (k[i] - 1) - pad[i] for i in range(3))
                dxc = _run_conv(dyc, packed, osc, Co, Ci, k, 1, p2, dil, precision, (D, H, W), dymeta)
            else:
                assert k == (3, 3, 3) and pad == (1, 1, 1) and dil == (1, 1, 1) and D % 2 == 0 and H % 2 == 0 and W % 2 == 0, \
                    "stride-2 data gradient: 3x3x3, pad 1, even input dims (what the aggregation networks use)"
                packed, osc = _pack(wf, Co, Ci, k, "deconv", precision, ctx.cache)       # w [Co][Ci][k] == transposed-conv layout [Cin_t][Cout_t]
                dxc = _run_deconv(dyc, packed, osc, Co, Ci, 3, 1, 1, precision)
            dx = (dxc[:, :Ci, 0] if ctx.flat else dxc[:, :Ci]).to(xdt)
        need_w = ctx.needs_input_grad[1] or (ctx.co1 is not None and ctx.needs_input_grad[7])
        need_b = ctx.bias_dt is not None and ctx.needs_input_grad[8]
        db = None
        if need_w or need_b:
            Do, Ho, Wo = dyc.shape[2:]

            def run(items):                                        # one weight-gradient launch (and one bias-gradient sum) over every queued (x, dy) pair of this weight
                from .ranges import combine_meta
                xl, dl = [it[0] for it in items], [it[1] for it in items]
                as_list = MULTI_WGRAD and 1 < len(items) <= 24 and _same_layout(xl) and _same_layout(dl)
                g = None
                if need_w:
                    xm = dm = None
                    if precision == "f16x3":
                        xm = combine_meta(*[it[2] if it[2] is not None else input_meta(it[0]) for it in items])
                        dm = combine_meta(*[it[3] for it in items])
                    g = torch.empty_like(wf)
                    if as_list:
                        _wgrad(xl, dl, g, len(items) * B, D, H, W, Ci, Do, Ho, Wo, Co, k, stride, pad, dil, 0, precision, xm, dm)
                    else:
                        xs, dl = _cat_batch(xl), [_cat_batch(dl)]
                        _wgrad(xs, dl[0], g, xs.shape[0], D, H, W, Ci, Do, Ho, Wo, Co, k, stride, pad, dil, 0, precision, xm, dm)
                return g, (_bias_grad_items(dl, Co) if need_b else None)
            dw, db = _defer_wgrad(ctx.cache, ("f32io", precision, tuple(xc.shape[1:]), tuple(dyc.shape[1:]), stride, pad, dil, need_w, need_b),
                                  (xc, dyc, ctx.xmeta, dymeta), run)
            if db is not None:
                db = db.to(ctx.bias_dt)
        if dw is not None and ctx.flat:
            dw = dw[:, :, 0]
        if ctx.co1 is not None:
            return dx, (None if dw is None else dw[:ctx.co1]), None, None, None, None, None, (None if dw is None else dw[ctx.co1:]), db, None
        return dx, dw, None, None, None, None, None, None, db, None


NATIVE_F16_IO = os.environ.get("OSA_NATIVE_F16_IO", "1") != "0"
IN_F16, OUT_F16 = 32, 64           # OSA_IN_F16 / OSA_OUT_F16 of include/openstereo_amd.h


def _cl16_ok(C):
    return C % 8 == 0


def _as_cl16(x):
    """logical [B,C,D,H,W] fp16 tensor -> the same values NDHWC-dense with a channel stride % 8 == 0 (zero-copy when it already is: outputs
    of this path, channels_last tensors; otherwise ONE torch copy -- no cast, no second layout pass).  Returns (tensor, channel stride)."""
    B, C, D, H, W = x.shape
    cs = x.stride(4) if W > 1 else (x.stride(3) if H > 1 else C)
    ok = x.stride(1) == 1 and cs >= C and cs % 8 == 0 and (x.data_ptr() % 16) == 0
    for dim, expect in ((4, cs), (3, W * cs), (2, H * W * cs), (0, D * H * W * cs)):
        ok = ok and (x.shape[dim] == 1 or x.stride(dim) == expect)
    if not ok:
        x = x.contiguous(memory_format=torch.channels_last_3d)
        cs = C
        if x.stride(1) != 1:                       # (degenerate shapes leave ambiguous strides)
            x = x.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
    return x, cs


class _Conv3dF16IO(torch.autograd.Function):
    """The stride-1 convolution in the native f16 mode with fp16 TENSORS (r5): what an AMP step of the reference moves between its layers.
    x arrives fp16 (any layout; NDHWC-dense is taken as it is), the result leaves fp16 when the reference's convolution would return fp16
    (`out_dtype`), the saved activation is the fp16 tensor, the data gradient is written fp16, the weight gradient reads fp16 x / dy
    (osa_conv3d_wgrad_ws_f16 x_f16 / dy_f16) and stays fp32 for the fp32 Parameter.  No cast kernels around the launch, half the saved
    bytes.  Same arithmetic as `_Conv3d` with precision "f16": operands are fp16 either way (there: rounded when they are staged)."""

    @staticmethod
    def forward(ctx, x, w, pad, dil, cache, w2, bias, flat, out_dtype):
        with torch.autocast("cuda", enabled=False):
            ctx.flat = flat
            if flat:
                x, w = x.unsqueeze(2), w.unsqueeze(2)
                w2 = None if w2 is None else w2.unsqueeze(2)
            xc, xcs = _as_cl16(x)
            ctx.co1 = None if w2 is None else w.shape[0]
            wf = _f32c(w) if w2 is None else _memo(cache, "wcat", lambda: torch.cat([_f32c(w.detach()), _f32c(w2.detach())], 0))
            Co, Ci = wf.shape[:2]
            k = tuple(wf.shape[2:])
            packed, _ = _pack(wf, Ci, Co, k, "fwd", "f16", cache)
            B, _, D, H, W = xc.shape
            oshape = (_out(D, k[0], pad[0], dil[0], 1), _out(H, k[1], pad[1], dil[1], 1), _out(W, k[2], pad[2], dil[2], 1))
            o16 = out_dtype == torch.float16 and _cl16_ok(Co)
            CoS = Co if o16 else (Co + 3) // 4 * 4
            y = empty_cl(B, CoS, *oshape, xc.device, torch.float16 if o16 else torch.float32)
            if CoS != Co:
                y.zero_()
            bf = None if bias is None else _f32c(bias.detach())
            ctx.bias_dt = None if bias is None else bias.dtype
            _launch_f16(xc, xcs, packed, y, CoS, [B, D, H, W, Ci], Co, k, pad, dil, IN_F16 | (OUT_F16 if o16 else 0), bf)
            ctx.save_for_backward(xc, wf)
            ctx.meta = (pad, dil, xcs, x.dtype)
            ctx.cache = cache
            _defer_note_use(cache, ctx.needs_input_grad[1] or (w2 is not None and ctx.needs_input_grad[5]) or (bias is not None and ctx.needs_input_grad[6]),
                            w, w2, flat, ctx.co1, bias)
            return _alias(y[:, :Co, 0] if flat else y[:, :Co])

    @staticmethod
    def backward(ctx, dy):
        with torch.autocast("cuda", enabled=False):
            xc, wf = ctx.saved_tensors
            pad, dil, xcs, xdt = ctx.meta
            Co, Ci = wf.shape[:2]
            k = tuple(wf.shape[2:])
            if ctx.flat:
                dy = dy.unsqueeze(2)
            if dy.dtype == torch.float16 and _cl16_ok(Co):
                dyc, dycs = _as_cl16(dy)
                fin = IN_F16
            else:
                dyc = to_cl(dy)
                dycs, fin = dyc.shape[1], 0
            B, _, D, H, W = xc.shape
            dx = dw = None
            if ctx.needs_input_grad[0]:
                packed, _ = _pack(wf, Co, Ci, k, "dgrad_s1", "f16", ctx.cache)
                p2 = tuple(dil[i] * (k[i] - 1) - pad[i] for i in range(3))
                dxc = empty_cl(B, Ci, D, H, W, xc.device, torch.float16)               # (Ci % 8 == 0: the entry condition of this path)
                cin = Co if fin else (Co + 3) // 4 * 4                                  # (fp32 dy from to_cl: zero-padded to a channel quad)
                _launch_f16(dyc, dycs, packed, dxc, Ci, [B, *dyc.shape[2:], cin], Ci, k, p2, dil, fin | OUT_F16, None)
                dx = (dxc[:, :, 0] if ctx.flat else dxc).to(xdt)
            need_w = ctx.needs_input_grad[1] or (ctx.co1 is not None and ctx.needs_input_grad[5])
            need_b = ctx.bias_dt is not None and ctx.needs_input_grad[6]
            db = None
            if need_w or need_b:
                Do, Ho, Wo = dyc.shape[2:]

                def run(items):                                    # one weight-gradient launch (and one bias-gradient sum) over every queued (x, dy) pair of this weight
                    xl, dl = [it[0] for it in items], [it[1] for it in items]
                    as_list = MULTI_WGRAD and 1 < len(items) <= 24 and _same_layout(xl) and _same_layout(dl)
                    g = None
                    if need_w:
                        g = torch.empty_like(wf)
                        if len(items) == 1:
                            _wgrad(xl[0], dl[0], g, B, D, H, W, Ci, Do, Ho, Wo, Co, k, 1, pad, dil, 0, "f16", xcs=xcs, dycs=dycs)
                        elif as_list:
                            _wgrad(xl, dl, g, len(items) * B, D, H, W, Ci, Do, Ho, Wo, Co, k, 1, pad, dil, 0, "f16", xcs=xcs, dycs=dycs)
                        else:                                      # (rows wider than the logical channels are dropped by the concatenation)
                            xs, dys = _cat_batch(xl, Ci), _cat_batch(dl, Co if fin else None)
                            _wgrad(xs, dys, g, xs.shape[0], D, H, W, Ci, Do, Ho, Wo, Co, k, 1, pad, dil, 0, "f16", xcs=xs.shape[1], dycs=dys.shape[1])
                    return g, (_bias_grad_items(dl, Co) if need_b else None)
                dw, db = _defer_wgrad(ctx.cache, ("f16io", tuple(xc.shape[1:]), tuple(dyc.shape[1:]), xcs, dycs, pad, dil, fin, need_w, need_b), (xc, dyc), run)
                if dw is not None and ctx.flat:
                    dw = dw[:, :, 0]
                if db is not None:
                    db = db.to(ctx.bias_dt)
            if ctx.co1 is not None:
                return dx, (None if dw is None else dw[:ctx.co1]), None, None, None, (None if dw is None else dw[ctx.co1:]), db, None, None
            return dx, dw, None, None, None, None, db, None, None


def _launch_f16(x, xcs, packed, y, ycs, xdims, Co, k, pad, dil, act, bias):
    """osa_conv3d_ndhwc_f16 (stride 1) on tensors of either float dtype (act carries OSA_IN_F16 / OSA_OUT_F16), bias in the epilogue"""
    B, D, H, W, Ci = xdims
    scale = None if bias is None else _ones(Co, x.device)
    dims = [B, D, H, W, Ci, xcs, Co, ycs, 0, 0]
    geom = [k[0], k[1], k[2], 1, pad[0], pad[1], pad[2], dil[0], dil[1], dil[2]]
    macs = B * y.shape[2] * y.shape[3] * y.shape[4] * Ci * Co * k[0] * k[1] * k[2]
    with timing.span("conv3d", Ci, Co, k[1], 1, D, H, W, flops=2 * macs, nbytes=x.element_size() * B * D * H * W * Ci + y.element_size() * y.numel()):
        ext = engine._ext.load()
        if ext is not None:
            ext.conv_ndhwc(x, 0, packed, scale, bias, None, 0, y, 0, None, dims, geom, 0, 2, act, 0.0, 1.0, [])
        else:
            _lib.call("osa_conv3d_ndhwc_f16", x.data_ptr(), packed.data_ptr(), _p(scale), _p(bias), None, y.data_ptr(),
                      B, D, H, W, Ci, xcs, Co, ycs, 0, k[0], k[1], k[2], 1, pad[0], pad[1], pad[2], dil[0], dil[1], dil[2], None, 0, act, 0.0, _stream())


def _native_f16(x, weight, stride, precision, dilation=1):
    """entry condition of the fp16-tensor path: the f16 mode, an fp16 input, unit stride / dilation and <= 3 taps per axis (what the fp16
    weight-gradient kernel covers: everything else keeps fp32 tensors and the exact weight gradient), input channels in whole 16-byte rows"""
    dl = dilation if isinstance(dilation, (tuple, list)) else (dilation,)
    return NATIVE_F16_IO and precision == "f16" and x.dtype == torch.float16 and x.is_cuda and stride == 1 and _cl16_ok(weight.shape[1]) \
        and all(int(v) <= 3 for v in weight.shape[2:]) and all(int(v) == 1 for v in dl)


class _ConvTranspose3d(torch.autograd.Function):
    """y = conv_transpose3d(x, w[Ci][Co][k], stride 2); k3/p1/op1 or k4/p1/op0."""

    @staticmethod
    @_fwd
    def forward(ctx, x, w, pad, opad, precision, cache=None):
        xc = to_cl(x)
        wf = _f32c(w)
        Ci, Co, k = wf.shape[0], wf.shape[1], wf.shape[2]
        packed, osc = _pack(wf, Ci, Co, (k, k, k), "deconv", precision, cache)
        ctx.cache = cache
        y = _run_deconv(xc, packed, osc, Ci, Co, k, pad, opad, precision)
        ctx.save_for_backward(xc, wf)
        ctx.meta = (pad, opad, precision, x.dtype)
        return _alias(y[:, :Co])

    @staticmethod
    @_bwd
    def backward(ctx, dy):
        xc, wf = ctx.saved_tensors
        pad, opad, precision, xdt = ctx.meta
        Ci, Co, k = wf.shape[0], wf.shape[1], wf.shape[2]
        dyc = to_cl(dy)
        B, _, D, H, W = xc.shape
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # dx = conv(dy, w) with stride 2: w read as [Cout'=Ci][Cin'=Co][k]
            packed, osc = _pack(wf, Co, Ci, (k, k, k), "dgrad_of_deconv", precision, ctx.cache)
            dxc = _run_conv(dyc, packed, osc, Co, Ci, (k, k, k), 2, (pad,) * 3, (1, 1, 1), precision, (D, H, W))
            dx = dxc[:, :Ci].to(xdt)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(wf)
            Do, Ho, Wo = dyc.shape[2:]
            _wgrad(xc, dyc, dw, B, D, H, W, Ci, Do, Ho, Wo, Co, (k, k, k), 2, (pad, pad, pad), (1, 1, 1), 1, precision)
        return dx, dw, None, None, None, None


class _ConvTranspose2d(torch.autograd.Function):
    """y = conv_transpose2d(x, w[Ci][Co][k][k], stride 2); k3/p1/op1 or k4/p1/op0: the D = 1 case (4 output-parity classes in one launch).
    The k = 4 upsampling heads of StereoBase / IGEV (spx_2_gru, spx_gru) run 22 times per training step; PyTorch-ROCm executes them as
    GEMM + col2im (5 ms per call at the 320x736 crop)."""

    @staticmethod
    @_fwd
    def forward(ctx, x, w, pad, opad, precision, cache=None):
        xc = to_cl(x.unsqueeze(2))
        ctx.cache = cache
        wf = _f32c(w)
        Ci, Co, k = wf.shape[0], wf.shape[1], wf.shape[2]
        packed, osc = _pack(wf, Ci, Co, (k, k), "deconv2d", precision, cache)
        B, Cs, _, H, W = xc.shape
        CoS = (Co + 3) // 4 * 4
        od = lambda n: (n - 1) * 2 - 2 * pad + k + opad
        y = empty_cl(B, CoS, 1, od(H), od(W), xc.device)
        if CoS != Co:
            y.zero_()
        Ci4 = (Ci + 3) // 4 * 4
        _defer_note_use(cache, ctx.needs_input_grad[1], w, None, True, None)      # (the k = 4 up-sampling heads run once per GRU iteration: batched weight gradient)
        if _ext_conv(2, xc, packed, y, [B, 1, H, W, Ci4, Cs, Co, CoS, 0, 0], [k, pad, opad], precision, osc):
            ctx.save_for_backward(xc, wf)
            ctx.meta = (pad, opad, precision, x.dtype)
            return _alias(y[:, :Co, 0])
        sfx, tail = _sfx_tail(precision, xc, y, osc)
        _lib.call("osa_deconv2d_nhwc_" + sfx, xc.data_ptr(), packed.data_ptr(), None, None, None, y.data_ptr(),
                  B, H, W, Ci4, Cs, Co, CoS, 0, k, pad, opad, None, 0, 0, 0.0, *tail)
        ctx.save_for_backward(xc, wf)
        ctx.meta = (pad, opad, precision, x.dtype)
        return _alias(y[:, :Co, 0])

    @staticmethod
    @_bwd
    def backward(ctx, dy):
        xc, wf = ctx.saved_tensors
        pad, opad, precision, xdt = ctx.meta
        Ci, Co, k = wf.shape[0], wf.shape[1], wf.shape[2]
        dyc = to_cl(dy.unsqueeze(2))
        B, _, _, H, W = xc.shape
        dx = dw = None
        w5 = wf.unsqueeze(2)                                               # [Ci][Co][1][k][k]
        if ctx.needs_input_grad[0]:
            packed, osc = _pack(w5, Co, Ci, (1, k, k), "dgrad_of_deconv", precision, ctx.cache)
            dxc = _run_conv(dyc, packed, osc, Co, Ci, (1, k, k), 2, (0, pad, pad), (1, 1, 1), precision, (1, H, W))
            dx = dxc[:, :Ci, 0].to(xdt)
        if ctx.needs_input_grad[1]:
            Ho, Wo = dyc.shape[3:]

            def run(items):                                        # one weight-gradient launch over every queued (x, dy) pair of this weight
                xl, dl = [it[0] for it in items], [it[1] for it in items]
                g = torch.empty_like(w5)
                if MULTI_WGRAD and 1 < len(items) <= 24 and _same_layout(xl) and _same_layout(dl):
                    _wgrad(xl, dl, g, len(items) * B, 1, H, W, Ci, 1, Ho, Wo, Co, (1, k, k), 2, (0, pad, pad), (1, 1, 1), 1, precision)
                else:
                    xs, dys = _cat_batch(xl), _cat_batch(dl)
                    _wgrad(xs, dys, g, xs.shape[0], 1, H, W, Ci, 1, Ho, Wo, Co, (1, k, k), 2, (0, pad, pad), (1, 1, 1), 1, precision)
                return g, None
            dw, _ = _defer_wgrad(ctx.cache, ("deconv2d", precision, tuple(xc.shape[1:]), tuple(dyc.shape[1:]), k, pad), (xc, dyc), run)
            if dw is not None:
                dw = dw[:, :, 0]
        return dx, dw, None, None, None, None


def _train_precision(precision):
    """arithmetic mode of the differentiable convolutions -- evaluated by the callers below BEFORE the Function runs (custom_fwd switches
    autocast off inside).  The requested / global mode; inside a `torch.autocast(fp16)` region the native "f16" mode (r5:
    engine.train_precision -- what the reference's AMP training computes, trainer_template.py:211-226; forward, data gradient and weight
    gradient then spend one MFMA per product instead of three and need no range reductions).  OSA_AMP_TRAIN_NATIVE=0 keeps f16x3 there."""
    return engine.train_precision(precision)


class _AddBias(torch.autograd.Function):
    """y + bias over the channel axis (the transposed convolutions add theirs outside the launch); the bias gradient is ONE pass of
    osa_channel_sums where the gradient's layout allows -- torch's sum_to_size of a channels-last gradient runs 17-97 us per call."""

    @staticmethod
    def forward(ctx, y, bias):
        ctx.bias_dt = bias.dtype
        return y + bias.view(1, -1, *([1] * (y.dim() - 2)))

    @staticmethod
    def backward(ctx, dy):
        db = None
        if ctx.needs_input_grad[1]:
            if cl_rows(dy) is not None:
                db = channel_sums(dy)[0][0].to(ctx.bias_dt)
            else:
                db = dy.sum([0] + list(range(2, dy.dim())), dtype=torch.float32).to(ctx.bias_dt)
        return dy, db


def conv_transpose2d(x, weight, bias=None, stride=2, padding=1, output_padding=0, precision=None):
    p2 = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    assert p2(stride) == (2, 2)
    y = _ConvTranspose2d.apply(x, weight, p2(padding)[0], p2(output_padding)[0], _train_precision(precision), _wcache(weight))
    return y if bias is None else _AddBias.apply(y, bias)


def _t3(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


def conv3d(x, weight, bias=None, stride=1, padding=0, dilation=1, precision=None):
    """Differentiable F.conv3d (groups=1, isotropic stride 1|2) on the engine; output is NDHWC-strided."""
    s = _t3(stride)
    assert s[0] == s[1] == s[2]
    prec = _train_precision(precision)
    if _native_f16(x, weight, s[0], prec, _t3(dilation)):
        return _Conv3dF16IO.apply(x, weight, _t3(padding), _t3(dilation), _wcache(weight), None, bias, False, amp.conv_out_dtype(x))
    return _Conv3d.apply(x, weight, s[0], _t3(padding), _t3(dilation), prec, _wcache(weight), None, bias)


def conv_transpose3d(x, weight, bias=None, stride=2, padding=1, output_padding=0, precision=None):
    assert _t3(stride) == (2, 2, 2)
    p, op = _t3(padding)[0], _t3(output_padding)[0]
    y = _ConvTranspose3d.apply(x, weight, p, op, _train_precision(precision), _wcache(weight))
    return y if bias is None else _AddBias.apply(y, bias)


def conv_module(m, x):
    """Run an nn.Conv3d / nn.ConvTranspose3d module's arithmetic on the engine (training path)."""
    if isinstance(m, torch.nn.ConvTranspose3d):
        return conv_transpose3d(x, m.weight, m.bias, m.stride, m.padding, m.output_padding)
    return conv3d(x, m.weight, m.bias, m.stride, m.padding, m.dilation)


# ----------------------------------------------------------------------------- 2-D convolutions and a module-level switch
def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, precision=None):
    """Differentiable F.conv2d (groups=1, stride 1) on the engine: the D = 1 case of conv3d (forward, dgrad and wgrad kernels)."""
    p2 = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    assert p2(stride) == (1, 1), "engine conv2d autograd: stride 1 (strided 2-D layers stay torch ops in training)"
    prec = _train_precision(precision)
    if _native_f16(x, weight, 1, prec, p2(dilation)):
        return _Conv3dF16IO.apply(x, weight, (0,) + p2(padding), (1,) + p2(dilation), _wcache(weight), None, bias, True, amp.conv_out_dtype(x))
    return _Conv3d.apply(x, weight, 1, (0,) + p2(padding), (1,) + p2(dilation), prec, _wcache(weight), None, bias, True)


def conv2d_pair(x, weight_a, weight_b, padding=0, dilation=1, precision=None):
    """[conv2d(x, weight_a) | conv2d(x, weight_b)] (no bias, stride 1) as ONE engine layer with the two weights stacked on the output axis:
    one forward, one data-gradient and one weight-gradient launch for both; the gradients come back per weight."""
    p2 = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    assert weight_a.shape[1:] == weight_b.shape[1:]
    prec = _train_precision(precision)
    if _native_f16(x, weight_a, 1, prec, p2(dilation)):
        return _Conv3dF16IO.apply(x, weight_a, (0,) + p2(padding), (1,) + p2(dilation), _wcache2(weight_a, weight_b), weight_b, None, True,
                                  amp.conv_out_dtype(x))
    return _Conv3d.apply(x, weight_a, 1, (0,) + p2(padding), (1,) + p2(dilation), prec, _wcache2(weight_a, weight_b), weight_b, None, True)


# ----------------------------------------------------------------------------- fused ConvGRU gates (training path, csrc/gru_train.hip)
def _nhwc_ref(t, C):
    """(tensor to keep alive, _lib.NhwcRef) for a logical [B, >=C, H, W] CUDA tensor read / written as NHWC with a channel stride: engine
    outputs (channel-sliced NDHWC views), channels_last tensors and anything whose strides are (H*W*cs, 1, W*cs, cs) pass as they are;
    other layouts / dtypes are converted once (differentiable torch ops; callers memoise per tensor object where an operand repeats)."""
    memo = getattr(t, "_osa_nhwc", None)            # operands that repeat (the context features cz / cr / cq: the same objects in all 22
    if memo is not None and memo[0] == t._version:  # GRU iterations of a step) are converted once
        t = memo[1]
    src = t
    if t.dtype not in (torch.float32, torch.float16):
        t = t.float()
    B, _, H, W = t.shape
    cs = t.stride(3)
    el = t.element_size()
    ok = t.stride(1) == 1 and cs >= C and cs % 4 == 0 and t.stride(2) == W * cs and t.stride(0) == H * W * cs and (t.data_ptr() % (4 * el)) == 0
    if not ok:
        t = t.contiguous(memory_format=torch.channels_last)
        if t.shape[1] % 4 or t.stride(1) != 1:          # (C % 4 == 0 is asserted by the callers; size-1 dims can leave ambiguous strides)
            t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        cs = t.stride(3)
    if t is not src:
        try:
            src._osa_nhwc = (src._version, t)
        except Exception:
            pass
    return t, _lib.NhwcRef(t.data_ptr(), int(cs), 1 if t.dtype == torch.float16 else 0)


def _new_nhwc(B, C, H, W, device, dtype):
    return torch.empty((B, H, W, C), device=device, dtype=dtype).permute(0, 3, 1, 2)


def _bias_ptr(b):
    return None if b is None else b.data_ptr()


class _GruGatesRZ(torch.autograd.Function):
    """z, r * h = sigmoid(pre_z + bz + cz), sigmoid(pre_r + br + cr) * h  (update.py:38-40), pre = [pre_z | pre_r] from conv2d_pair."""

    @staticmethod
    def forward(ctx, pre, bz, br, cz, cr, h, rh_dtype):
        B, C2, H, W = pre.shape
        C = C2 // 2
        keep = [_nhwc_ref(t, n) for t, n in ((pre, C2), (cz, C), (cr, C), (h, C))]
        bzf, brf = (None if bz is None else _f32c(bz.detach())), (None if br is None else _f32c(br.detach()))
        z = _new_nhwc(B, C, H, W, pre.device, torch.float32)             # internal: only the q kernel reads it
        rh = _new_nhwc(B, C, H, W, pre.device, rh_dtype)
        ext = engine._ext.load()
        if ext is not None:
            ext.gru_gates_rz_fwd(keep[0][0], bzf, brf, keep[1][0], keep[2][0], keep[3][0], z, rh)
        else:
            zr, rhr = _nhwc_ref(z, C), _nhwc_ref(rh, C)
            _lib.call("osa_gru_gates_rz_fwd", keep[0][1], _bias_ptr(bzf), _bias_ptr(brf), keep[1][1], keep[2][1], keep[3][1], zr[1], rhr[1],
                      B * H * W, C, _stream())
        ctx.save_for_backward(keep[0][0], keep[1][0], keep[2][0], keep[3][0], *([] if bzf is None else [bzf]), *([] if brf is None else [brf]))
        ctx.has_b = (bzf is not None, brf is not None)
        ctx.dt = (pre.dtype, cz.dtype, cr.dtype, h.dtype, None if bz is None else bz.dtype, None if br is None else br.dtype)
        return z, rh

    @staticmethod
    def backward(ctx, dz, drh):
        sv = list(ctx.saved_tensors)
        pre, cz, cr, h = sv[:4]
        rest = sv[4:]
        bzf = rest.pop(0) if ctx.has_b[0] else None
        brf = rest.pop(0) if ctx.has_b[1] else None
        B, C2, H, W = pre.shape
        C = C2 // 2
        refs = [_nhwc_ref(t, n) for t, n in ((pre, C2), (cz, C), (cr, C), (h, C), (dz, C), (drh, C))]
        dpre = _new_nhwc(B, C2, H, W, pre.device, torch.float32)
        dh = _new_nhwc(B, C, H, W, pre.device, torch.float32)
        ext = engine._ext.load()
        if ext is not None:
            ext.gru_gates_rz_bwd(refs[0][0], bzf, brf, refs[1][0], refs[2][0], refs[3][0], refs[4][0], refs[5][0], dpre, dh)
        else:
            _lib.call("osa_gru_gates_rz_bwd", refs[0][1], _bias_ptr(bzf), _bias_ptr(brf), refs[1][1], refs[2][1], refs[3][1], refs[4][1], refs[5][1],
                      _nhwc_ref(dpre, C2)[1], _nhwc_ref(dh, C)[1], B * H * W, C, _stream())
        pdt, czdt, crdt, hdt, bzdt, brdt = ctx.dt
        db = _bias_grad(dpre.unsqueeze(2), C2) if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else None       # one reduction for both biases
        return (dpre.to(pdt), None if bzdt is None or db is None else db[:C].to(bzdt), None if brdt is None or db is None else db[C:].to(brdt),
                dpre[:, :C].to(czdt) if ctx.needs_input_grad[3] else None, dpre[:, C:].to(crdt) if ctx.needs_input_grad[4] else None,
                dh.to(hdt) if ctx.needs_input_grad[5] else None, None)


class _GruGatesQ(torch.autograd.Function):
    """h' = (1 - z) * h + z * tanh(qpre + bq + cq)  (update.py:41-44)"""

    @staticmethod
    def forward(ctx, z, qpre, bq, cq, h, out_dtype):
        B, C, H, W = qpre.shape
        keep = [_nhwc_ref(t, C) for t in (z, qpre, cq, h)]
        bqf = None if bq is None else _f32c(bq.detach())
        out = _new_nhwc(B, C, H, W, qpre.device, out_dtype)
        ext = engine._ext.load()
        if ext is not None:
            ext.gru_gates_q_fwd(keep[0][0], keep[1][0], bqf, keep[2][0], keep[3][0], out)
        else:
            _lib.call("osa_gru_gates_q_fwd", keep[0][1], keep[1][1], _bias_ptr(bqf), keep[2][1], keep[3][1], _nhwc_ref(out, C)[1], B * H * W, C, _stream())
        ctx.save_for_backward(keep[0][0], keep[1][0], keep[2][0], keep[3][0], *([] if bqf is None else [bqf]))
        ctx.dt = (z.dtype, qpre.dtype, cq.dtype, h.dtype, None if bq is None else bq.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        sv = list(ctx.saved_tensors)
        z, qpre, cq, h = sv[:4]
        bqf = sv[4] if len(sv) > 4 else None
        B, C, H, W = qpre.shape
        refs = [_nhwc_ref(t, C) for t in (z, qpre, cq, h, dout)]
        dz, dq, dh = (_new_nhwc(B, C, H, W, qpre.device, torch.float32) for _ in range(3))
        ext = engine._ext.load()
        if ext is not None:
            ext.gru_gates_q_bwd(refs[0][0], refs[1][0], bqf, refs[2][0], refs[3][0], refs[4][0], dz, dq, dh)
        else:
            _lib.call("osa_gru_gates_q_bwd", refs[0][1], refs[1][1], _bias_ptr(bqf), refs[2][1], refs[3][1], refs[4][1],
                      _nhwc_ref(dz, C)[1], _nhwc_ref(dq, C)[1], _nhwc_ref(dh, C)[1], B * H * W, C, _stream())
        zdt, qdt, cqdt, hdt, bqdt = ctx.dt
        return (dz.to(zdt), dq.to(qdt), None if bqdt is None or not ctx.needs_input_grad[2] else _bias_grad(dq.unsqueeze(2), C).to(bqdt),
                dq.to(cqdt) if ctx.needs_input_grad[3] else None, dh.to(hdt) if ctx.needs_input_grad[4] else None, None)


def gru_gates_rz(pre, bias_z, bias_r, cz, cr, h, rh_dtype=None):
    with torch.autocast("cuda", enabled=False):
        return _GruGatesRZ.apply(pre, bias_z, bias_r, cz, cr, h, rh_dtype or h.dtype)


def gru_gates_q(z, qpre, bias_q, cq, h, out_dtype=None):
    with torch.autocast("cuda", enabled=False):
        return _GruGatesQ.apply(z, qpre, bias_q, cq, h, out_dtype or h.dtype)


def _wgrad_ok(m, x):
    """Preconditions of the backward kernels for module m on input x, checked BEFORE the forward is routed to the engine (a
    dilated 3x3 conv whose weight-gradient brick exceeds the 160 KB of LDS used to pass `_eligible` and raise inside backward()).  Only
    consulted when a gradient can actually be asked for."""
    if not (torch.is_grad_enabled() and (m.weight.requires_grad or x.requires_grad)):
        return True
    nn = torch.nn
    tr = isinstance(m, (nn.ConvTranspose2d, nn.ConvTranspose3d))
    flat = isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))
    lift = (lambda v, unit: (unit,) + tuple(v)) if flat else (lambda v, unit: tuple(v))      # 2-D layer = the D = 1 case
    k, s, p, d = lift(m.kernel_size, 1), lift(m.stride, 1), lift(m.padding, 0), lift(m.dilation, 1)
    sp = ((1,) + tuple(x.shape[2:])) if flat else tuple(x.shape[2:])
    if len(sp) != 3 or any(v <= 0 for v in sp):
        return False
    st = max(s)
    if tr:
        op = lift(m.output_padding, 0)
        out = tuple((sp[i] - 1) * s[i] - 2 * p[i] + d[i] * (k[i] - 1) + op[i] + 1 for i in range(3))
        Ci, Co = m.in_channels, m.out_channels
    else:
        out = tuple((sp[i] + 2 * p[i] - d[i] * (k[i] - 1) - 1) // s[i] + 1 for i in range(3))
        Ci, Co = m.in_channels, m.out_channels
        if st == 2 and not flat and any(v % 2 for v in sp):            # stride-2 data gradient runs as the fused transposed conv: even dims
            return False
    if any(v <= 0 for v in out):
        return False
    need = _lib.load().osa_conv3d_wgrad_workspace_bytes(int(x.shape[0]), sp[0], sp[1], sp[2], Ci, out[0], out[1], out[2], Co,
                                                        k[0], k[1], k[2], st, p[0], p[1], p[2], d[0], d[1], d[2], 1 if tr else 0)
    return need != 0


def _eligible(m, x):
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype in (torch.float32, torch.float16, torch.bfloat16)):
        return False
    if m.groups != 1 or getattr(m, "padding_mode", "zeros") != "zeros" or isinstance(m.padding, str):
        return False
    return _shape_eligible(m, x) and _wgrad_ok(m, x)


def _shape_eligible(m, x):
    if isinstance(m, torch.nn.Conv2d):
        return tuple(m.stride) == (1, 1) and m.in_channels >= 4
    if isinstance(m, torch.nn.ConvTranspose2d):
        k, p, op = m.kernel_size, m.padding, m.output_padding
        return tuple(m.stride) == (2, 2) and tuple(m.dilation) == (1, 1) and m.in_channels >= 4 and \
            ((tuple(k), tuple(p), tuple(op)) in (((3, 3), (1, 1), (1, 1)), ((4, 4), (1, 1), (0, 0))))
    if isinstance(m, torch.nn.ConvTranspose3d):
        k, p, op = m.kernel_size, m.padding, m.output_padding
        return tuple(m.stride) == (2, 2, 2) and tuple(m.dilation) == (1, 1, 1) and \
            ((tuple(k), tuple(p), tuple(op)) in (((3, 3, 3), (1, 1, 1), (1, 1, 1)), ((4, 4, 4), (1, 1, 1), (0, 0, 0))))
    if isinstance(m, torch.nn.Conv3d):
        s = tuple(m.stride)
        if s == (1, 1, 1):
            return True
        return s == (2, 2, 2) and tuple(m.kernel_size) == (3, 3, 3) and tuple(m.padding) == (1, 1, 1) and tuple(m.dilation) == (1, 1, 1) \
            and all(d % 2 == 0 for d in x.shape[2:])
    return False



# ----------------------------------------------------------------------------- fused softmax + convex up-sampling (training)
class _ContextUpsampleLogits(torch.autograd.Function):
    """out[B,H,W] = sum_k softmax(logits)_k * (gain * disp_low)[3x3 neighbourhood]_k  (stereobase_gru.py:196-203: F.softmax(spx_gru(..), 1) +
    context_upsample(disp * 4, spx_pred), once per GRU iteration for the sequence loss).  One kernel forward, two backward
    (osa_context_upsample_logits_f32 / _bwd_f32); the logits are read in the layout and dtype the transposed conv wrote them."""

    @staticmethod
    def forward(ctx, disp_low, logits, scale, gain):
        d = _f32c(disp_low)
        lg = logits if logits.dtype in (torch.float16, torch.float32) else logits.float()
        B, _, h, w = d.shape
        ext = engine._ext.load()
        if ext is not None:
            out = ext.context_upsample_logits(d, lg, scale, gain)
        else:
            out = torch.empty((B, h * scale, w * scale), device=d.device, dtype=torch.float32)
            _lib.call("osa_context_upsample_logits_f32", d.data_ptr(), lg.data_ptr(), int(lg.dtype == torch.float16), (ctypes.c_longlong * 4)(*lg.stride()),
                      out.data_ptr(), B, h, w, scale, float(gain), _stream())
        ctx.save_for_backward(d, lg)
        ctx.meta = (scale, gain, disp_low.dtype, logits.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        d, lg = ctx.saved_tensors
        scale, gain, ddt, ldt = ctx.meta
        B, _, h, w = d.shape
        g = _f32c(dout)
        ext = engine._ext.load()
        if ext is not None:
            dd, dl = ext.context_upsample_logits_bwd(d, lg, g, scale, gain)
        else:
            dd = torch.empty_like(d)
            dl = torch.empty(lg.shape, device=lg.device, dtype=lg.dtype)      # contiguous NCHW, as the torch composition's softmax backward returns it
            sc = torch.empty((B, 9, h, w), device=d.device, dtype=torch.float32)
            _lib.call("osa_context_upsample_logits_bwd_f32", d.data_ptr(), lg.data_ptr(), int(lg.dtype == torch.float16), (ctypes.c_longlong * 4)(*lg.stride()),
                      g.data_ptr(), dd.data_ptr(), dl.data_ptr(), (ctypes.c_longlong * 4)(*dl.stride()), sc.data_ptr(), B, h, w, scale, float(gain), _stream())
        return (dd.to(ddt) if ctx.needs_input_grad[0] else None), (dl.to(ldt) if ctx.needs_input_grad[1] else None), None, None


def context_upsample_logits(disp_low, logits, scale_factor=4, gain=1.0):
    """differentiable `context_upsample(disp_low * gain, F.softmax(logits, 1))` -> [B, H, W] fp32"""
    with torch.autocast("cuda", enabled=False):
        return _ContextUpsampleLogits.apply(disp_low, logits, int(scale_factor), float(gain))


# ----------------------------------------------------------------------------- BatchNorm in eval mode (FREEZE_BN training)
class _FrozenBN(torch.autograd.Function):
    """y = batch_norm(x) with the running statistics (a BatchNorm module in eval mode whose affine parameters still train: the reference's
    FREEZE_BN, trainer_template.py:83-85).  Forward is torch's own inference kernel (bit-identical to the unpatched module); backward is ONE
    pass of osa_channel_sums over (dy, x): dbeta = sum dy, dgamma = invstd * sum dy (x - mean), dx = dy * gamma * invstd -- torch runs a
    channels-last reduce kernel plus two elementwise kernels for it (7 % of the StereoBase AMP step)."""

    @staticmethod
    def forward(ctx, x, weight, bias, mean, var, eps):
        y = torch.nn.functional.batch_norm(x, mean, var, weight, bias, False, 0.0, eps)
        ctx.save_for_backward(x, weight, mean, var)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, var = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        if dy.dtype != x.dtype or cl_rows(dy) is None or cl_rows(x) is None or dy.shape != x.shape:
            # other layouts (a contiguous NCHW gradient from a torch op): the same three formulas as torch ops
            with torch.autocast("cuda", enabled=False):
                bc = [1, -1] + [1] * (x.dim() - 2)
                red = [0] + list(range(2, x.dim()))
                invstd = torch.rsqrt(var.float() + ctx.eps)
                g = dy.float()
                dx = (g * (w.float() * invstd).view(bc)).to(dy.dtype) if need_x else None
                dg = ((g * (x.float() - mean.float().view(bc))).sum(red) * invstd).to(w.dtype) if need_w else None
                db = g.sum(red).to(w.dtype) if need_b else None
            return dx, dg, db, None, None, None
        with torch.autocast("cuda", enabled=False):
            invstd = torch.rsqrt(var.float() + ctx.eps)
            sums, dx = channel_sums(dy, x if need_w else None, mean if need_w else None, (w.float() * invstd) if need_x else None)
            dg = (sums[1] * invstd).to(w.dtype) if need_w else None
            db = sums[0].to(w.dtype) if need_b else None
        return dx, dg, db, None, None, None


class _TrainBN(torch.autograd.Function):
    """BatchNorm in TRAINING mode (batch statistics) on channels-last tensors: GwcNet / PSMNet train this way (gwcnet_disp_processor.py:8-19
    convbn_3d + nn.BatchNorm3d in train(); cfgs/gwcnet/gwcnet_sceneflow.yaml).  Forward: one osa_channel_sums pass (sum x, sum x (x - pivot),
    pivot = the running mean: no cancellation once it tracks) -> mean / biased variance -> one osa_channel_affine pass; backward: one
    sums pass (sum dy, sum dy (x - mean)) and one affine pass (dx = dy a + x b + c).  Running statistics are updated as nn.BatchNorm does
    (momentum, unbiased variance).  Not SyncBatchNorm (its forward is its own; statistics across ranks stay torch's)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        P, C, _ = cl_rows(x)
        pivot = running_mean.detach().float()
        sums, _ = channel_sums(x, x, pivot)
        s1 = sums[0]
        mean = s1 / P
        dm = mean - pivot
        var = ((sums[1] - pivot * (s1 - P * pivot)) / P - dm * dm).clamp_min_(0.0)          # sum (x - p)^2 / N - (mean - p)^2
        invstd = torch.rsqrt(var + eps)
        g = weight.detach().float()
        a = g * invstd
        y = ops.channel_affine(x, a, bias.detach().float() - mean * a)
        with torch.no_grad():
            running_mean.mul_(1 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
            running_var.mul_(1 - momentum).add_((var * (P / max(P - 1, 1))).to(running_var.dtype), alpha=momentum)
        ctx.save_for_backward(x, g, mean, invstd)
        ctx.wdt = weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, invstd = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        if dy.dtype != x.dtype or cl_rows(dy) is None or dy.shape != x.shape:
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last if x.dim() == 4 else torch.channels_last_3d)
            if cl_rows(dy) is None:
                raise _lib.EngineError("engine BatchNorm backward: gradient layout not representable as channels-last rows")
        P, C, _ = cl_rows(x)
        with torch.autocast("cuda", enabled=False):
            sums, _ = channel_sums(dy, x, mean)
            s1, s2 = sums[0], sums[1]
            dx = None
            if need_x:
                a = g * invstd
                b = -(a * invstd * invstd) * (s2 / P)
                dx = ops.channel_affine(dy, a, -a * (s1 / P) - b * mean, x, b)
            return dx, ((s2 * invstd).to(ctx.wdt) if need_w else None), (s1.to(ctx.wdt) if need_b else None), None, None, None, None


def _train_bn_ok(m, x):
    return m.training and not isinstance(m, torch.nn.SyncBatchNorm) and m.track_running_stats and m.affine and m.momentum is not None and m.running_mean is not None \
        and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() in (4, 5) and cl_rows(x) is not None \
        and m.weight.dtype == torch.float32 and m.running_mean.dtype == torch.float32 and x.numel() // x.shape[1] > 1


def _frozen_bn_ok(m, x):
    return (not m.training) and m.track_running_stats and m.affine and m.running_mean is not None and isinstance(x, torch.Tensor) and x.is_cuda \
        and torch.is_grad_enabled() and (x.requires_grad or m.weight.requires_grad) and x.dim() in (4, 5) and cl_rows(x) is not None \
        and m.weight.dtype == torch.float32 and m.running_mean.dtype == torch.float32


def bn_module(m, x):
    """A BatchNorm module's arithmetic on the engine where it applies (the same dispatch as inside `engine_convs`): eval mode with
    trainable affine parameters -> _FrozenBN, training mode -> _TrainBN, anything else -> the module itself."""
    if FROZEN_BN and _frozen_bn_ok(m, x):
        return _FrozenBN.apply(x, m.weight, m.bias, m.running_mean, m.running_var, m.eps)
    if TRAIN_BN and _train_bn_ok(m, x):
        if m.num_batches_tracked is not None:
            m.num_batches_tracked.add_(1)
        return _TrainBN.apply(x, m.weight, m.bias, m.running_mean, m.running_var, float(m.momentum), m.eps)
    return m(x)


class engine_convs:
    """Context manager: inside it every eligible nn.Conv3d / nn.ConvTranspose3d / nn.Conv2d forward -- and its backward -- runs on the
    engine's kernels through the autograd Functions above; BatchNorm, activations, depthwise / strided 2-D convolutions, pooling and
    everything else stay the torch modules they are, so batch statistics, SyncBN and DistributedDataParallel behave exactly as in the
    reference.  This is the training path of modules that keep the reference's own forward (attach: train_fallback) and of the mirrors'
    forward_train methods:

        with openstereo_amd.autograd.engine_convs():
            loss = model(batch)            # reference-built or mirror model, in train mode
        loss.backward()                    # backward kernels run outside the context too (they belong to the recorded Functions)
    """
    _depth = 0                      # number of threads x nesting levels currently inside (guarded by _lock)
    _saved = {}
    _lock = threading.RLock()
    _tls = threading.local()        # .depth: nesting level of THIS thread; the patched forwards reroute only when it is > 0

    @classmethod
    def active(cls) -> bool:
        return getattr(cls._tls, "depth", 0) > 0

    def __enter__(self):
        cls = engine_convs
        with cls._lock:
            if cls._depth == 0:
                nn = torch.nn
                BN = nn.modules.batchnorm._BatchNorm
                for C in (nn.Conv2d, nn.Conv3d, nn.ConvTranspose3d, nn.ConvTranspose2d, BN, nn.SyncBatchNorm):
                    cls._saved[C] = C.forward
                obn, osbn = cls._saved[BN], cls._saved[nn.SyncBatchNorm]

                def fsbn(m, x):             # SyncBatchNorm has a forward of its own: only its eval-mode (frozen) case is taken -- the same affine map as BatchNorm's
                    if on() and FROZEN_BN and _frozen_bn_ok(m, x):
                        return _FrozenBN.apply(x, m.weight, m.bias, m.running_mean, m.running_var, m.eps)
                    return osbn(m, x)
                nn.SyncBatchNorm.forward = fsbn

                def fbn(m, x):
                    if on() and FROZEN_BN and _frozen_bn_ok(m, x):
                        return _FrozenBN.apply(x, m.weight, m.bias, m.running_mean, m.running_var, m.eps)
                    if on() and TRAIN_BN and _train_bn_ok(m, x):
                        if m.num_batches_tracked is not None:
                            m.num_batches_tracked.add_(1)
                        return _TrainBN.apply(x, m.weight, m.bias, m.running_mean, m.running_var, float(m.momentum), m.eps)
                    return obn(m, x)
                BN.forward = fbn
                o2, o3, ot, ot2 = cls._saved[nn.Conv2d], cls._saved[nn.Conv3d], cls._saved[nn.ConvTranspose3d], cls._saved[nn.ConvTranspose2d]
                on = cls.active             # other threads (a validation thread, DataParallel replicas) keep the original forwards

                def ft2(m, x, output_size=None):
                    if on() and output_size is None and _eligible(m, x):
                        return conv_transpose2d(x, m.weight, m.bias, m.stride, m.padding, m.output_padding).to(amp.conv_out_dtype(x))
                    return ot2(m, x, output_size)

                def f2(m, x):
                    return conv2d(x, m.weight, m.bias, m.stride, m.padding, m.dilation).to(amp.conv_out_dtype(x)) if (on() and _eligible(m, x)) else o2(m, x)

                def f3(m, x):
                    return conv_module(m, x).to(amp.conv_out_dtype(x)) if (on() and _eligible(m, x)) else o3(m, x)

                def ft(m, x, output_size=None):
                    return conv_module(m, x).to(amp.conv_out_dtype(x)) if (on() and output_size is None and _eligible(m, x)) else ot(m, x, output_size)
                nn.Conv2d.forward, nn.Conv3d.forward, nn.ConvTranspose3d.forward, nn.ConvTranspose2d.forward = f2, f3, ft, ft2
            cls._depth += 1
        cls._tls.depth = getattr(cls._tls, "depth", 0) + 1
        return self

    def __exit__(self, *exc):
        cls = engine_convs
        cls._tls.depth = getattr(cls._tls, "depth", 1) - 1
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0:         # the last thread out restores the class attributes
                for C, f in cls._saved.items():
                    C.forward = f
                cls._saved.clear()
        return False
