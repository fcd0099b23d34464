"""`torch.ops.openstereo_amd.*`: the hot-path entry points registered with the PyTorch dispatcher (what an
extension replacing the reference's helpers must export).

Each op is a `torch.library.custom_op` whose CUDA (= ROCm) implementation calls the C-ABI library, with
  * a fake / meta kernel (shape + dtype inference), so `torch.compile`, `torch.export` and FakeTensor tracing -- e.g. the
    reference's deploy/export.py -- see an opaque op instead of failing on a ctypes call,
  * an autograd formula that runs the engine's backward kernels (osa_build_volume_bwd_f32, osa_*softargmin*_bwd_f32),
  * autocast behaviour of the eager composition it replaces: fp16 / bf16 in -> same dtype out, arithmetic in fp32.
There is no CPU kernel: calling an op on CPU tensors raises NotImplementedError from the dispatcher (no fallback).

    import openstereo_amd.torch_ops            # registers the ops
    vol = torch.ops.openstereo_amd.gwc_volume(left, right, 48, 40)

Reference interfaces: cost_volume.py:32-41,59-92 (volumes), disp_regression.py:8-12 and gwcnet_disp_processor.py:22-26,
128-133 (regression heads), disp_refinement.py:194-204 (context_upsample).
"""
from __future__ import annotations

import torch
from torch.library import custom_op

from . import _lib, ops
from .ops import _f32c, _stream

NS = "openstereo_amd"


# ----------------------------------------------------------------------------- volumes
@custom_op(f"{NS}::gwc_volume", mutates_args=(), device_types="cuda")
def gwc_volume(left: torch.Tensor, right: torch.Tensor, maxdisp: int, num_groups: int) -> torch.Tensor:
    return ops.build_gwc_volume(left, right, maxdisp, num_groups)


@gwc_volume.register_fake
def _(left, right, maxdisp, num_groups):
    torch._check(left.dim() == 4 and left.shape == right.shape, lambda: "gwc_volume: [B,C,H,W] feature maps of equal shape")
    torch._check(left.shape[1] % num_groups == 0, lambda: "gwc_volume: C % num_groups != 0")          # cost_volume.py:61
    B, _, H, W = left.shape
    return left.new_empty((B, num_groups, maxdisp, H, W))


def _vol_bwd(ctx, dvol, concat, mask_left):
    l, r = ctx.saved_tensors
    B, C, H, W = ctx.shape
    dv = _f32c(dvol)
    dl = torch.empty((B, C, H, W), device=dv.device, dtype=torch.float32)
    dr = torch.empty_like(dl)
    if concat:
        _lib.call("osa_build_volume_bwd_f32", dv.data_ptr(), None, None, dl.data_ptr(), dr.data_ptr(),
                  B, C, H, W, ctx.maxdisp, 0, 1, 1 if mask_left else 0, 2 * C, 0, _stream())
    else:
        _lib.call("osa_build_volume_bwd_f32", dv.data_ptr(), l.data_ptr(), r.data_ptr(), dl.data_ptr(), dr.data_ptr(),
                  B, C, H, W, ctx.maxdisp, ctx.groups, 0, 1, ctx.groups, 0, _stream())
    return dl.to(ctx.dtype), dr.to(ctx.dtype)


def _gwc_setup(ctx, inputs, output):
    left, right, maxdisp, groups = inputs
    ctx.save_for_backward(_f32c(left), _f32c(right))
    ctx.shape, ctx.maxdisp, ctx.groups, ctx.dtype = tuple(left.shape), maxdisp, groups, left.dtype


gwc_volume.register_autograd(lambda ctx, g: (*_vol_bwd(ctx, g, False, True), None, None), setup_context=_gwc_setup)


@custom_op(f"{NS}::concat_volume", mutates_args=(), device_types="cuda")
def concat_volume(left: torch.Tensor, right: torch.Tensor, maxdisp: int, mask_left: bool = True) -> torch.Tensor:
    return ops.build_concat_volume(left, right, maxdisp, mask_left=mask_left)


@concat_volume.register_fake
def _(left, right, maxdisp, mask_left=True):
    B, C, H, W = left.shape
    return left.new_empty((B, 2 * C, maxdisp, H, W))


def _cat_setup(ctx, inputs, output):
    left, right, maxdisp, mask_left = inputs
    ctx.save_for_backward(left.new_empty(0), left.new_empty(0))
    ctx.shape, ctx.maxdisp, ctx.mask_left, ctx.dtype = tuple(left.shape), maxdisp, mask_left, left.dtype


concat_volume.register_autograd(lambda ctx, g: (*_vol_bwd(ctx, g, True, ctx.mask_left), None, None), setup_context=_cat_setup)


@custom_op(f"{NS}::corr_volume", mutates_args=(), device_types="cuda")
def corr_volume(left: torch.Tensor, right: torch.Tensor, maxdisp: int) -> torch.Tensor:
    return ops.correlation_volume(left, right, maxdisp)


@corr_volume.register_fake
def _(left, right, maxdisp):
    B, _, H, W = left.shape
    return left.new_empty((B, maxdisp, H, W))


def _corr_setup(ctx, inputs, output):
    left, right, maxdisp = inputs
    ctx.save_for_backward(_f32c(left), _f32c(right))
    ctx.shape, ctx.maxdisp, ctx.groups, ctx.dtype = tuple(left.shape), maxdisp, 1, left.dtype


corr_volume.register_autograd(lambda ctx, g: (*_vol_bwd(ctx, g.unsqueeze(1), False, True), None), setup_context=_corr_setup)


# ----------------------------------------------------------------------------- regression heads
@custom_op(f"{NS}::softargmin", mutates_args=(), device_types="cuda")
def softargmin(prob: torch.Tensor, keepdim: bool = True) -> torch.Tensor:
    return ops.disparity_regression(prob, prob.shape[1], keepdim)


@softargmin.register_fake
def _(prob, keepdim=True):
    B, D, H, W = prob.shape
    return prob.new_empty((B, 1, H, W) if keepdim else (B, H, W))


def _sa_setup(ctx, inputs, output):
    ctx.shape, ctx.dtype = tuple(inputs[0].shape), inputs[0].dtype


def _sa_bwd(ctx, g):
    B, D, H, W = ctx.shape
    gg = _f32c(g.reshape(B, H, W))
    dp = torch.empty(ctx.shape, device=gg.device, dtype=torch.float32)
    _lib.call("osa_softargmin_bwd_f32", gg.data_ptr(), dp.data_ptr(), B, D, H, W, _stream())
    return dp.to(ctx.dtype), None


softargmin.register_autograd(_sa_bwd, setup_context=_sa_setup)


@custom_op(f"{NS}::softmax_softargmin", mutates_args=(), device_types="cuda")
def softmax_softargmin(cost: torch.Tensor, keepdim: bool = True) -> torch.Tensor:
    out = ops.softmax_disparity_regression(cost, keepdim=keepdim)
    return out if cost.dtype == torch.float32 else out.to(cost.dtype)


@softmax_softargmin.register_fake
def _(cost, keepdim=True):
    B, D, H, W = cost.shape
    return cost.new_empty((B, 1, H, W) if keepdim else (B, H, W))


def _ssa_setup(ctx, inputs, output):
    ctx.save_for_backward(_f32c(inputs[0]))
    ctx.dtype = inputs[0].dtype


def _ssa_bwd(ctx, g):
    (c,) = ctx.saved_tensors
    B, D, H, W = c.shape
    gg = _f32c(g.reshape(B, H, W))
    dc = torch.empty_like(c)
    _lib.call("osa_softmax_softargmin_bwd_f32", c.data_ptr(), gg.data_ptr(), dc.data_ptr(), B, D, H, W, _stream())
    return dc.to(ctx.dtype), None


softmax_softargmin.register_autograd(_ssa_bwd, setup_context=_ssa_setup)


@custom_op(f"{NS}::upsample_softargmin", mutates_args=(), device_types="cuda")
def upsample_softargmin(cost_lowres: torch.Tensor, maxdisp: int, h: int, w: int, align_corners: bool = False) -> torch.Tensor:
    out = ops.upsample_softargmin(cost_lowres, maxdisp, h, w, align_corners)
    return out if cost_lowres.dtype == torch.float32 else out.to(cost_lowres.dtype)


@upsample_softargmin.register_fake
def _(cost_lowres, maxdisp, h, w, align_corners=False):
    torch._check(cost_lowres.dim() in (4, 5), lambda: "upsample_softargmin: [B,Dl,Hl,Wl] or [B,1,Dl,Hl,Wl]")
    return cost_lowres.new_empty((cost_lowres.shape[0], h, w))


def _usa_setup(ctx, inputs, output):
    c, maxdisp, h, w, align = inputs
    ctx.five = c.dim() == 5
    ctx.save_for_backward(_f32c(c[:, 0] if ctx.five else c))
    ctx.meta, ctx.dtype = (maxdisp, h, w, align), c.dtype


def _usa_bwd(ctx, g):
    (c,) = ctx.saved_tensors
    maxdisp, h, w, align = ctx.meta
    B, Dl, Hl, Wl = c.shape
    gg = _f32c(g)
    dc = torch.empty_like(c)
    need = _lib.load().osa_upsample_softargmin_bwd_workspace_bytes(B, Dl, int(h), int(w))
    ws = torch.empty((need + 3) // 4, device=c.device, dtype=torch.float32)
    _lib.call("osa_upsample_softargmin_bwd_ws_f32", c.data_ptr(), gg.data_ptr(), dc.data_ptr(), B, Dl, Hl, Wl,
              int(maxdisp), int(h), int(w), 1 if align else 0, ws.data_ptr(), need, _stream())
    dc = dc.to(ctx.dtype)
    return (dc.unsqueeze(1) if ctx.five else dc), None, None, None, None


upsample_softargmin.register_autograd(_usa_bwd, setup_context=_usa_setup)


# ----------------------------------------------------------------------------- refinement
@custom_op(f"{NS}::context_upsample", mutates_args=(), device_types="cuda")
def context_upsample(disp_low: torch.Tensor, up_weights: torch.Tensor, scale_factor: int = 4, softmax_weights: bool = False,
                     gain: float = 1.0) -> torch.Tensor:
    return ops.context_upsample(disp_low, up_weights, scale_factor, softmax_weights, gain)


@context_upsample.register_fake
def _(disp_low, up_weights, scale_factor=4, softmax_weights=False, gain=1.0):
    b, _, h, w = disp_low.shape
    return disp_low.new_empty((b, h * scale_factor, w * scale_factor))


OPS = ("gwc_volume", "concat_volume", "corr_volume", "softargmin", "softmax_softargmin", "upsample_softargmin", "context_upsample")
