"""Python mirror of OpenStereo's hot-path *functions* on top of the gfx950 C-ABI library.

Same names, argument meaning, output shapes/dtypes and error behaviour as the reference
helpers; every call lands in a hand-written HIP kernel.  torch is used for device
memory and the current stream only.

Reference                                                      here
  cost_volume.build_gwc_volume(ref,tgt,maxdisp,groups)         build_gwc_volume
  cost_volume.build_concat_volume(ref,tgt,maxdisp)             build_concat_volume
  cost_volume.correlation_volume / build_corr_volume           correlation_volume / build_corr_volume
  psmnet_cost_processor.cat_fms                                cat_fms
  igev.submodule.build_concat_volume (left half unmasked)      build_concat_volume(..., mask_left=False)
  disp_pred.disparity_regression (keepdim=True)                disparity_regression
  gwcnet_disp_processor.disparity_regression (keepdim=False)   disparity_regression(..., keepdim=False)
  psmnet_disp_processor.FasterSoftArgmin                       FasterSoftArgmin
Engine-only fused entry points (no reference equivalent, they replace op chains):
  build_cost_volume_cl, softmax_disparity_regression, upsample_softargmin
"""
from __future__ import annotations

import torch

from . import _ext, _lib, timing
from .ranges import attach_meta

NCDHW, NDHWC = 0, 1


def on_engine(t) -> bool:
    """True when tensor `t` lives where the engine runs (a GPU).  The one place the module forwards ask; the CPU
    wiring tests (tests/engine_emulation.py) swap it together with the engine layer classes."""
    return t.is_cuda


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str, ndim: int | None = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise _lib.EngineError(f"{name} is on {t.device}: the gfx950 engine has no CPU path")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name} must be {ndim}-D, got shape {tuple(t.shape)}")
    return t


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32 contiguous view/copy (kernels compute in fp32; callers cast back)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


def _sum_dtype(*ts):
    """dtype of `torch.sum(a * b ...)` as the reference writes it (disp_regression.py:11, disp_refinement.py:203): the promoted input
    dtype -- except inside a CUDA autocast region, where `sum` is on the fp32 list (openstereo_amd/amp.py)."""
    if torch.is_autocast_enabled("cuda"):
        return torch.float32
    dt = ts[0].dtype
    for t in ts[1:]:
        dt = torch.promote_types(dt, t.dtype)
    return dt


# --------------------------------------------------------------------------- layouts
def empty_cl(B, C, D, H, W, device, dtype=torch.float32) -> torch.Tensor:
    """Logical [B,C,D,H,W] tensor stored NDHWC (torch.channels_last_3d strides)."""
    return torch.empty((B, D, H, W, C), device=device, dtype=dtype).permute(0, 4, 1, 2, 3)


def is_cl(t: torch.Tensor) -> bool:
    return t.dim() == 5 and t.permute(0, 2, 3, 4, 1).is_contiguous()


def to_cl(x: torch.Tensor, pad_to: int = 4) -> torch.Tensor:
    """NCDHW -> NDHWC through the engine's transpose kernel.  Channel count is padded up to a
    multiple of `pad_to` (zero filled) because conv inputs are read as float4."""
    _chk(x, "x", 5)
    if is_cl(x) and x.shape[1] % pad_to == 0 and x.dtype == torch.float32:
        return x
    xs = _f32c(x)
    B, Cn, D, H, W = xs.shape
    Cp = (Cn + pad_to - 1) // pad_to * pad_to
    y = empty_cl(B, Cp, D, H, W, x.device)
    if Cp != Cn:
        y.zero_()
    ext = _ext.load()
    if ext is not None:
        ext.to_cl(xs, y, Cn, D * H * W, 0)
    else:
        _lib.call("osa_ncdhw_to_ndhwc_f32", xs.data_ptr(), y.data_ptr(), B, Cn, D * H * W, Cp, 0, _stream())
    return y


def to_ncdhw(x: torch.Tensor, channels: int | None = None) -> torch.Tensor:
    """NDHWC -> contiguous NCDHW (first `channels` channels)."""
    assert not getattr(x, "_osa_split", False), "split activation tensors are internal to engine layer chains"
    _chk(x, "x", 5)
    if not is_cl(x):
        return x.contiguous()
    if x.dtype != torch.float32:         # fp16 chain tensors of the f16 mode: the layout kernel counts the channel stride in floats
        raise _lib.EngineError(f"to_ncdhw takes an fp32 NDHWC tensor, got {x.dtype} (an f16-mode chain tensor is internal to its layer chain)")
    B, Cs, D, H, W = x.shape
    Cn = Cs if channels is None else channels
    y = torch.empty((B, Cn, D, H, W), device=x.device, dtype=torch.float32)
    ext = _ext.load()
    if ext is not None:
        ext.to_ncdhw(x, y, Cn, D * H * W, 0)
    else:
        _lib.call("osa_ndhwc_to_ncdhw_f32", x.data_ptr(), y.data_ptr(), B, Cn, D * H * W, Cs, 0, _stream())
    return y


# --------------------------------------------------------------------------- volumes
def _build(lg, rg, G, lc, rc, maxdisp, layout, mask_left=True, out=None, vol_channels=None, c_off=0):
    ref = lg if lg is not None else lc
    B, _, H, W = ref.shape
    Cg = lg.shape[1] if lg is not None else 0
    Cc = lc.shape[1] if lc is not None else 0
    nch = (G if Cg else 0) + 2 * Cc
    VC = nch if vol_channels is None else vol_channels
    if out is None:
        if layout == NDHWC:
            out = empty_cl(B, VC, maxdisp, H, W, ref.device)
            if VC > c_off + nch or c_off > 0:
                out.zero_()
        else:
            out = torch.empty((B, VC, maxdisp, H, W), device=ref.device, dtype=torch.float32)
    meta = attach_meta(out) if layout == NDHWC else None     # range block for f16x3 consumers
    with timing.span("build_volume", Cg, G, Cc, layout, maxdisp, H, W):
        ext = _ext.load()
        if ext is not None:
            ext.build_volume(lg, rg, G, lc, rc, out, layout, VC, c_off, maxdisp, bool(mask_left), meta)
        else:
            _lib.call("osa_build_volume_f32", _p(lg), _p(rg), Cg, G, _p(lc), _p(rc), Cc,
                      out.data_ptr(), layout, VC, c_off, B, H, W, maxdisp, 1 if mask_left else 0, _p(meta), _stream())
    return out


def build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """cost_volume.py:68-78 -> [B, num_groups, maxdisp, H, W], contiguous, input dtype."""
    _chk(refimg_fea, "refimg_fea", 4); _chk(targetimg_fea, "targetimg_fea", 4)
    B, Cn, H, W = refimg_fea.shape
    assert Cn % num_groups == 0                                   # cost_volume.py:61
    v = _build(_f32c(refimg_fea), _f32c(targetimg_fea), num_groups, None, None, maxdisp, NCDHW)
    return v if refimg_fea.dtype == torch.float32 else v.to(refimg_fea.dtype)


def build_concat_volume(refimg_fea, targetimg_fea, maxdisp, mask_left=True):
    """cost_volume.py:81-92 -> [B, 2C, maxdisp, H, W]. mask_left=False is IGEV's copy (submodule.py:216-227)."""
    _chk(refimg_fea, "refimg_fea", 4); _chk(targetimg_fea, "targetimg_fea", 4)
    v = _build(None, None, 0, _f32c(refimg_fea), _f32c(targetimg_fea), maxdisp, NCDHW, mask_left=mask_left)
    return v if refimg_fea.dtype == torch.float32 else v.to(refimg_fea.dtype)


def correlation_volume(left_feature, right_feature, max_disp):
    """cost_volume.py:32-41 -> [B, max_disp, H, W] (mean over all channels)."""
    _chk(left_feature, "left_feature", 4); _chk(right_feature, "right_feature", 4)
    l, r = _f32c(left_feature), _f32c(right_feature)
    B, Cn, H, W = l.shape
    ext = _ext.load()
    if ext is not None:
        out = ext.corr_volume(l, r, int(max_disp))
    else:
        out = torch.empty((B, max_disp, H, W), device=l.device, dtype=torch.float32)
        _lib.call("osa_corr_volume_f32", l.data_ptr(), r.data_ptr(), out.data_ptr(), B, Cn, H, W, max_disp, _stream())
    return out if left_feature.dtype == torch.float32 else out.to(left_feature.dtype)


def build_corr_volume(img_left, img_right, max_disp):
    """cost_volume.py:95-105: correlation_volume, except planes d >= W repeat the d=0 plane
    (the reference's `(i > 0) & (i < W)` guard sends them to the unshifted branch)."""
    vol = correlation_volume(img_left, img_right, max_disp)
    W = img_left.shape[-1]
    if max_disp > W:
        vol[:, W:] = vol[:, :1]
    return vol


def cat_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1):
    """psmnet_cost_processor.py:9-50; always returns fp32 like the reference (its buffer is torch.zeros(...).to(device)).  The
    configuration PSMNet uses (start_disp=0, dilation=1) is the fused concat-volume kernel; any other sampling (negative start, dilation)
    runs osa_cat_fms_f32 on the reference's own index list int(torch.linspace(start, end, n))."""
    ref, tgt = _f32c(_chk(reference_fm, "reference_fm", 4)), _f32c(_chk(target_fm, "target_fm", 4))
    if start_disp == 0 and dilation == 1:
        return _build(None, None, 0, ref, tgt, max_disp, NCDHW)
    B, C, H, W = ref.shape
    n = (max_disp + dilation - 1) // dilation                                          # psmnet_cost_processor.py:31-33
    idx = torch.tensor([int(i) for i in torch.linspace(start_disp, start_disp + max_disp - 1, n)], dtype=torch.int32, device=ref.device)
    out = torch.empty((B, 2 * C, n, H, W), device=ref.device, dtype=torch.float32)
    ext = _ext.load()
    if ext is not None:
        ext.cat_fms(ref, tgt, out, idx)
    else:
        _lib.call("osa_cat_fms_f32", ref.data_ptr(), tgt.data_ptr(), out.data_ptr(), idx.data_ptr(), B, C, H, W, n, _stream())
    return out


def _pair_volume(left, right, planes, mode, groups=1):
    l, r = _f32c(_chk(left, "left", 4)), _f32c(_chk(right, "right", 4))
    assert l.shape == r.shape, "feature maps must have the same shape"
    B, C, H, W = l.shape
    shape = (B, groups, planes, H, W) if mode == 0 else ((B, planes, H, W) if mode == 3 else (B, C, planes, H, W))
    out = torch.empty(shape, device=l.device, dtype=torch.float32)
    ext = _ext.load()
    if ext is not None:
        ext.pair_volume(l, r, out, groups, planes, mode)
    else:
        _lib.call("osa_pair_volume_f32", l.data_ptr(), r.data_ptr(), out.data_ptr(), B, C, groups, H, W, planes, mode, _stream())
    return out if left.dtype == torch.float32 else out.to(left.dtype)


def compute_volume(reference_embedding, target_embedding, maxdisp, side="left"):
    """cost_volume.py:44-56 (difference volume, [B,C,maxdisp,H,W])."""
    assert side in ("left", "right")
    return _pair_volume(reference_embedding, target_embedding, maxdisp, 1 if side == "left" else 2)


def build_sub_volume(feat_l, feat_r, maxdisp):
    """cost_volume.py:108-117 (L1 volume, [B,maxdisp,H,W])."""
    return _pair_volume(feat_l, feat_r, maxdisp, 3)


class CoExCostVolume(torch.nn.Module):
    """cost_volume.py:9-29: correlation summed over the channels of each group, maxdisp + 1 planes, [B,group,maxdisp+1,H,W]."""

    def __init__(self, maxdisp, group=1):
        super().__init__()
        self.maxdisp, self.group = maxdisp + 1, group

    def forward(self, x, y):
        return _pair_volume(x, y, self.maxdisp, 0, self.group)


def build_cost_volume_cl(gwc_left, gwc_right, num_groups, cat_left=None, cat_right=None, maxdisp=48,
                         mask_left=True):
    """Fused gwc(+concat) volume written straight into one NDHWC buffer
    (replaces build_gwc_volume + build_concat_volume + torch.cat, gwcnet_cost_processor.py:55-68).
    Returns logical [B, G+2Cc (padded to 4), D, H, W] with channels_last_3d strides."""
    lg = rg = None
    if gwc_left is not None:
        lg, rg = _f32c(_chk(gwc_left, "gwc_left", 4)), _f32c(_chk(gwc_right, "gwc_right", 4))
        assert lg.shape[1] % num_groups == 0
    else:
        num_groups = 0
    lc = rc = None
    Cc = 0
    if cat_left is not None:
        lc, rc = _f32c(_chk(cat_left, "cat_left", 4)), _f32c(_chk(cat_right, "cat_right", 4))
        Cc = lc.shape[1]
    nch = num_groups + 2 * Cc
    VC = (nch + 3) // 4 * 4
    return _build(lg, rg, num_groups, lc, rc, maxdisp, NDHWC, mask_left=mask_left, vol_channels=VC)


def build_cost_volume_from_cl(gwc_feat, num_groups, cat_feat, B, maxdisp, gwc_channels=None, cat_channels=None,
                              gwc_off=0, mask_left=True, out_split=False):
    """Volume from the engine backbone's channels-last feature maps.  gwc_feat / cat_feat: logical
    [2B, Cs, 1, H, W] NDHWC tensors holding the left images first ([0:B]) and the right images last.
    out_split=True (f16x3 chains): when the call qualifies (osa_build_volume_nhwc_split_eligible) and both feature tensors carry range
    blocks, the volume is written as a split tensor (tagged `_osa_split`, scale from the features' ranges) for the aggregation chain;
    otherwise the fp32 volume is returned as always -- the consumer recognises the format by the tag."""
    assert is_cl(gwc_feat) and gwc_feat.shape[0] == 2 * B and gwc_feat.shape[2] == 1
    _, Gs, _, H, W = gwc_feat.shape
    C = Gs - gwc_off if gwc_channels is None else gwc_channels
    img = H * W * Gs * 4
    Cc = cs = 0
    if cat_feat is not None:
        assert is_cl(cat_feat) and cat_feat.shape[0] == 2 * B and tuple(cat_feat.shape[3:]) == (H, W)
        cs = cat_feat.shape[1]
        Cc = cs if cat_channels is None else cat_channels
    nch = num_groups + 2 * Cc
    VC = (nch + 3) // 4 * 4
    gm = getattr(gwc_feat, "_osa_meta", None) if C > 0 else None
    cm = getattr(cat_feat, "_osa_meta", None) if (cat_feat is not None and Cc > 0) else None
    ext = _ext.load()
    if ext is not None and gwc_feat.dtype == torch.float32 and (cat_feat is None or cat_feat.dtype == torch.float32):
        # PyTorch-ROCm C++ extension (csrc/torch_ext.cpp cost_volume_cl): allocation, eligibility check and launch in one dispatcher call
        from .ranges import new_meta
        om = new_meta(gwc_feat.device)
        with timing.span("build_volume", C, num_groups, Cc, NDHWC, maxdisp, H, W):
            out, split = ext.cost_volume_cl(gwc_feat, cat_feat, B, num_groups, maxdisp, C, Cc, gwc_off, bool(mask_left), bool(out_split), gm, cm, om)
        out._osa_meta = om
        if split:
            out._osa_split = True
        return out
    # ctypes path: raw addresses
    lg, rg = gwc_feat.data_ptr() + 4 * gwc_off, gwc_feat.data_ptr() + 4 * gwc_off + B * img
    lc = rc = None
    if cat_feat is not None:
        lc, rc = cat_feat.data_ptr(), cat_feat.data_ptr() + B * H * W * cs * 4
    out = empty_cl(B, VC, maxdisp, H, W, gwc_feat.device)
    if VC != nch:
        out.zero_()
    split = bool(out_split) and VC == nch and (C == 0 or gm is not None) and (Cc == 0 or cm is not None) and \
        _lib.load().osa_build_volume_nhwc_split_eligible(lc, rc, out.data_ptr(), C, num_groups, Gs, Cc, cs, VC, 0, W, maxdisp) == 1
    with timing.span("build_volume", C, num_groups, Cc, NDHWC, maxdisp, H, W):
        if split:
            _lib.call("osa_build_volume_nhwc_split_f16x3", lg, rg, C, num_groups, Gs, lc, rc, Cc, cs, out.data_ptr(), VC, 0,
                      B, H, W, maxdisp, 1 if mask_left else 0, None if gm is None else gm.data_ptr(), None if cm is None else cm.data_ptr(),
                      attach_meta(out).data_ptr(), _stream())
            out._osa_split = True
        else:
            _lib.call("osa_build_volume_nhwc_f32", lg, rg, C, num_groups, Gs, lc, rc, Cc, cs, out.data_ptr(), VC, 0,
                      B, H, W, maxdisp, 1 if mask_left else 0, attach_meta(out).data_ptr(), _stream())
    return out


# --------------------------------------------------------------------------- regression
def disparity_regression(x, maxdisp, keepdim=True):
    """disp_regression.py:8-12 (keepdim=True) / gwcnet_disp_processor.py:22-26 (keepdim=False)."""
    assert len(x.shape) == 4                                      # disp_regression.py:9
    _chk(x, "x")
    B, D, H, W = x.shape
    assert D == maxdisp, f"x has {D} disparity planes, maxdisp={maxdisp}"
    xs = _f32c(x)
    ext = _ext.load()
    if ext is not None:                                           # torch extension: at::Tensor in / out (csrc/torch_ext.cpp)
        out = ext.softargmin(xs)
    else:
        out = torch.empty((B, H, W), device=x.device, dtype=torch.float32)
        _lib.call("osa_softargmin_f32", xs.data_ptr(), out.data_ptr(), B, D, H, W, _stream())
    out = out if _sum_dtype(x) == torch.float32 else out.to(x.dtype)
    return out.unsqueeze(1) if keepdim else out


def softmax_disparity_regression(cost, maxdisp=None, keepdim=True, return_prob=False):
    """F.softmax(cost, dim=1) + disparity_regression in one kernel (stereobase_gru.py:163-164)."""
    assert len(cost.shape) == 4
    _chk(cost, "cost")
    B, D, H, W = cost.shape
    if maxdisp is not None:
        assert D == maxdisp
    cs = _f32c(cost)
    ext = _ext.load()
    if ext is not None:
        out, prob = ext.softmax_softargmin(cs, bool(return_prob))
    else:
        out = torch.empty((B, H, W), device=cost.device, dtype=torch.float32)
        prob = torch.empty_like(cs) if return_prob else None
        _lib.call("osa_softmax_softargmin_f32", cs.data_ptr(), _p(prob), out.data_ptr(), B, D, H, W, _stream())
    out = out.unsqueeze(1) if keepdim else out
    return (out, prob) if return_prob else out


def upsample_softargmin(cost_lowres, maxdisp, h, w, align_corners=False):
    """F.interpolate(cost[:,None], [maxdisp,h,w], 'trilinear') -> squeeze -> softmax(dim=1) ->
    disparity_regression(keepdim=False), fused (gwcnet_disp_processor.py:128-133; PSMNet uses
    align_corners=True, psmnet_cost_processor.py:201-214).  cost_lowres: [B,Dl,Hl,Wl] or [B,1,Dl,Hl,Wl]."""
    _chk(cost_lowres, "cost_lowres")
    if cost_lowres.dim() == 5:
        assert cost_lowres.shape[1] == 1
        cost_lowres = cost_lowres[:, 0]
    assert cost_lowres.dim() == 4
    cs = _f32c(cost_lowres)
    B, Dl, Hl, Wl = cs.shape
    ext = _ext.load()
    with timing.span("upsample_softargmin", Dl, Hl, Wl, int(maxdisp), int(h), int(w)):
        if ext is not None:
            return ext.upsample_softargmin(cs, int(maxdisp), int(h), int(w), bool(align_corners))
        out = torch.empty((B, h, w), device=cs.device, dtype=torch.float32)
        _lib.call("osa_upsample_softargmin_f32", cs.data_ptr(), out.data_ptr(), B, Dl, Hl, Wl,
                  int(maxdisp), int(h), int(w), 1 if align_corners else 0, _stream())
    return out


def context_upsample(disp_low, up_weights, scale_factor=4, softmax_weights=False, gain=1.0):
    """disp_refinement.py:194-204 / stereobase/igev_blocks.py:51-63 / igev/submodule.py:253-265:
    disp_low [b,1,h,w], up_weights [b,9,s*h,s*w] -> [b,s*h,s*w].  softmax_weights / gain fuse the
    F.softmax(spx_pred, 1) and `disp * 4.` the callers apply first (lightstereo.py:61-62)."""
    _chk(disp_low, "disp_low", 4); _chk(up_weights, "up_weights", 4)
    b, c, h, w = disp_low.shape
    assert c == 1 and tuple(up_weights.shape) == (b, 9, h * scale_factor, w * scale_factor)
    d, wt = _f32c(disp_low), _f32c(up_weights)
    ext = _ext.load()
    with timing.span("context_upsample", h, w, scale_factor):
        if ext is not None:
            out = ext.context_upsample(d, wt, int(scale_factor), bool(softmax_weights), float(gain))
        else:
            out = torch.empty((b, h * scale_factor, w * scale_factor), device=d.device, dtype=torch.float32)
            _lib.call("osa_context_upsample_f32", d.data_ptr(), wt.data_ptr(), out.data_ptr(), b, h, w, int(scale_factor),
                      1 if softmax_weights else 0, float(gain), _stream())
    od = _sum_dtype(disp_low, up_weights)
    return out if od == torch.float32 else out.to(od)


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def preprocess_pair(left_hwc, right_hwc, pad_size, mean=IMAGENET_MEAN, std=IMAGENET_STD, channels_last=False):
    """RightTopPad(SIZE=pad_size, edge) -> TransposeImage -> ToTensor -> NormalizeImage for a stereo pair,
    fused on the GPU (stereo_trans.py:243-267,22-29,48-56).  left/right: [H,W,3] uint8 or float32 CUDA tensors.
    Returns (left, right) [1,3,Hp,Wp]; channels_last=True returns the engine's NHWC4 pair tensor [2,4,1,Hp,Wp]."""
    import ctypes
    _chk(left_hwc, "left_hwc", 3); _chk(right_hwc, "right_hwc", 3)
    assert left_hwc.shape == right_hwc.shape and left_hwc.shape[2] == 3 and left_hwc.dtype == right_hwc.dtype
    assert left_hwc.dtype in (torch.uint8, torch.float32)
    H, W = left_hwc.shape[:2]
    Hp, Wp = max(pad_size[0], H), max(pad_size[1], W)        # RightTopPad never crops (h = min(h, th))
    l, r = left_hwc.contiguous(), right_hwc.contiguous()
    m3 = (ctypes.c_float * 3)(*mean); s3 = (ctypes.c_float * 3)(*std)
    if channels_last:
        out = torch.empty((2, 1, Hp, Wp, 4), device=l.device, dtype=torch.float32)
    else:
        out = torch.empty((2, 3, Hp, Wp), device=l.device, dtype=torch.float32)
    ext = _ext.load()
    if ext is not None:
        ext.preprocess_pair(l, r, out, [Hp, Wp], [float(v) for v in mean], [float(v) for v in std], bool(channels_last))
    else:
        _lib.call("osa_preprocess_pair_f32", l.data_ptr(), r.data_ptr(), 1 if l.dtype == torch.uint8 else 0, H, W, Hp, Wp,
                  m3, s3, out.data_ptr(), 1 if channels_last else 0, _stream())
    if channels_last:
        return out.permute(0, 4, 1, 2, 3)
    return out[0:1], out[1:2]


class FasterSoftArgmin(torch.nn.Module):
    """psmnet_disp_processor.py:6-74: softmax over D + expectation; keeps the frozen
    `disp_regression.weight` buffer name so PSMNet checkpoints load."""

    def __init__(self, max_disp, start_disp=0, dilation=1, alpha=1.0, normalize=True):
        super().__init__()
        self.max_disp, self.start_disp, self.dilation = max_disp, start_disp, dilation
        self.end_disp = start_disp + max_disp - 1
        self.disp_sample_number = (max_disp + dilation - 1) // dilation
        self.alpha, self.normalize = alpha, normalize
        self.disp_regression = torch.nn.Conv3d(1, 1, (self.disp_sample_number, 1, 1), 1, 0, bias=False)
        with torch.no_grad():
            self.disp_regression.weight.copy_(
                torch.linspace(self.start_disp, self.end_disp, self.disp_sample_number).view(1, 1, -1, 1, 1))
        self.disp_regression.weight.requires_grad = False

    def forward(self, cost_volume):
        if cost_volume.dim() != 4:                                # psmnet_disp_processor.py:56-58
            raise ValueError('expected 4D input (got {}D input)'.format(cost_volume.dim()))
        if self.start_disp != 0 or self.dilation != 1:
            raise NotImplementedError("FasterSoftArgmin: only start_disp=0, dilation=1")
        c = cost_volume * self.alpha if self.alpha != 1.0 else cost_volume
        if torch.is_grad_enabled() and c.requires_grad:          # training: forward + backward on the engine through autograd
            from . import autograd as AG
            out = AG.softmax_disparity_regression(c, keepdim=False) if self.normalize else AG.disparity_regression(c, c.shape[1], keepdim=False)
            return out.to(cost_volume.dtype)
        if self.normalize:
            return softmax_disparity_regression(c, keepdim=False)
        return disparity_regression(c, c.shape[1], keepdim=False)


def cl_rows(t: torch.Tensor):
    """(P, C, cs) when `t` (logical [B, C, *spatial], fp32 / fp16) is a dense run of P channel rows of stride cs elements -- engine
    outputs and their channel slices, channels_last / channels_last_3d tensors -- with the alignment the vector kernels need; else None."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dim() >= 3 and t.dtype in (torch.float32, torch.float16)):
        return None
    C = t.shape[1]
    if C > 1 and t.stride(1) != 1:
        return None
    dims = [0] + list(range(2, t.dim()))                 # batch, then the spatial dims, outermost first
    cs, expect = None, None
    for d in reversed(dims):
        if t.shape[d] == 1:
            continue
        if cs is None:
            cs = expect = t.stride(d)
        elif t.stride(d) != expect:
            return None
        expect = expect * t.shape[d]
    if cs is None:
        cs = (C + 3) // 4 * 4 if C % 4 else C
    if cs < C or cs % 4 or t.data_ptr() % (8 if t.dtype == torch.float16 else 16):
        return None
    return t.numel() // C, C, cs


def channel_sums(dy: torch.Tensor, x: torch.Tensor | None = None, x_shift: torch.Tensor | None = None, dx_scale: torch.Tensor | None = None):
    """Per-channel sums over the positions of a channels-last tensor (osa_channel_sums): returns (sums, dx) with sums[0] = sum_p dy,
    sums[1] = sum_p dy * (x - x_shift) when x is given, dx = dy * dx_scale[c] when dx_scale is given (else None).  Bias gradient of a
    convolution, backward of an eval-mode BatchNorm.  dy / x must satisfy `cl_rows` (the caller checks; anything else is its torch path)."""
    P, C, cs = cl_rows(dy)
    xcs = 0
    if x is not None:
        Px, Cx, xcs = cl_rows(x)
        assert (Px, Cx) == (P, C), "channel_sums: x and dy must agree in shape"
    sh = None if x_shift is None else _f32c(x_shift)
    sc = None if dx_scale is None else _f32c(dx_scale)
    ext = _ext.load()
    if ext is not None:
        out, dx = ext.channel_sums(dy, x, sh, sc, P, C, cs, xcs)
        return out, (dx if sc is not None else None)
    lib = _lib.load()
    need = lib.osa_channel_sums_workspace_bytes(P, C)
    if not need:
        raise _lib.EngineError(f"osa_channel_sums: unsupported dims P={P} C={C}")
    out = torch.empty((2 if x is not None else 1, C), device=dy.device, dtype=torch.float32)
    ws = torch.empty((need + 3) // 4, device=dy.device, dtype=torch.float32)
    dx = torch.empty_strided(dy.shape, dy.stride(), device=dy.device, dtype=dy.dtype) if sc is not None else None
    _lib.call("osa_channel_sums", dy.data_ptr(), int(dy.dtype == torch.float16), cs, _p(x), int(x is not None and x.dtype == torch.float16), xcs,
              _p(sh), _p(sc), _p(dx), cs, P, C, out.data_ptr(), ws.data_ptr(), need, _stream())
    return out, dx


def channel_sums_list(dys):
    """sum_p dy over a list of equally shaped and strided tensors (osa_channel_sums_multi): [C] fp32, one launch, no concatenation"""
    P, C, cs = cl_rows(dys[0])
    ext = _ext.load()
    if ext is not None:
        return ext.channel_sums_multi(list(dys), P, C, cs)[0]
    import ctypes
    lib = _lib.load()
    need = len(dys) * lib.osa_channel_sums_workspace_bytes(P, C)
    if not need:
        raise _lib.EngineError(f"osa_channel_sums_multi: unsupported dims P={P} C={C}")
    out = torch.empty((1, C), device=dys[0].device, dtype=torch.float32)
    ws = torch.empty((need + 3) // 4, device=dys[0].device, dtype=torch.float32)
    _lib.call("osa_channel_sums_multi", (ctypes.c_void_p * len(dys))(*[t.data_ptr() for t in dys]), len(dys), int(dys[0].dtype == torch.float16), cs, P, C,
              out.data_ptr(), ws.data_ptr(), need, _stream())
    return out[0]


def channel_affine(u, a, c0, v=None, b=None, relu=False):
    """u * a[c] + (v * b[c]) + c0[c] (+ ReLU) on channels-last rows (osa_channel_affine); u / v must satisfy `cl_rows`; result: u's dtype / strides"""
    P, C, cs = cl_rows(u)
    vcs = 0
    if v is not None:
        Pv, Cv, vcs = cl_rows(v)
        assert (Pv, Cv) == (P, C) and b is not None
    a, c0 = _f32c(a), _f32c(c0)
    b = None if b is None else _f32c(b)
    ext = _ext.load()
    if ext is not None:
        return ext.channel_affine(u, v, a, b, c0, P, C, cs, vcs, bool(relu))
    out = torch.empty_strided(u.shape, u.stride(), device=u.device, dtype=u.dtype)
    _lib.call("osa_channel_affine", u.data_ptr(), int(u.dtype == torch.float16), cs, _p(v), int(v is not None and v.dtype == torch.float16), vcs,
              a.data_ptr(), _p(b), c0.data_ptr(), out.data_ptr(), cs, P, C, int(bool(relu)), _stream())
    return out
