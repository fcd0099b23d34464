"""3-D aggregation layers on the gfx950 engine: packed convolutions with fused BN/activation/residual.

A PackedConv3d is built once from the *reference-shaped* torch parameters (nn.Conv3d /
nn.ConvTranspose3d weight + eval-mode BatchNorm3d statistics), so state_dict keys and shapes stay
exactly the reference's (SURVEY 8b "checkpoint compatibility"); only forward() changes.
Activations travel NDHWC (torch.channels_last_3d strides) between layers.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _ext, _lib, timing
from .ops import _stream, _p, empty_cl, is_cl

import math
import os

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_RELU6, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4, 5
GATE_RAW = 16          # OR'ed into act: the gate is a plain multiplier, not sigmoid logits
# "split" activation tensors (f16x3 mode): every 16-channel chunk stored as [16 x fp16 hi | 16 x fp16 lo]
# (same bytes as fp32), see include/openstereo_amd.h.  A tensor written that way carries `_osa_split = True`.
IN_SPLIT, OUT_SPLIT, RES_SPLIT, REDIR_SPLIT = 32, 64, 128, 256
RES_AFTER_ACT = 512    # OR'ed into act (ReLU layers): relu(residual + relu(bn(conv))) -- MultiBasicEncoder's ResidualBlock


def is_split(t) -> bool:
    """tensor is in an engine-chain storage format: f16x3 split hi/lo (tagged `_osa_split`) or an fp16 tensor of the f16 mode"""
    return t is not None and (getattr(t, "_osa_split", False) or t.dtype == torch.float16)


def chains(precision: str) -> bool:
    """layers of this arithmetic mode hand each other chain-format tensors (`out_split=True`): split hi/lo in f16x3, fp16 in f16"""
    return precision in ("f16x3", "f16")


def chain_ok(layer) -> bool:
    """`layer`'s output can be written in its mode's chain format: every 16-channel chunk complete (f16x3) / 8-channel row complete (f16)"""
    return (layer.precision == "f16x3" and layer.Co % 16 == 0) or (layer.precision == "f16" and layer.Co % 8 == 0)


from .ranges import META_FLOATS, new_meta, meta_of, input_meta, ensure_meta, fold_amax, attach_meta   # noqa: E402,F401  (f16x3 operand ranges)


enable_timing, collect_timing = timing.enable, timing.collect

# Arithmetic mode of the MFMA convolutions:
#   "f32"   exact fp32 products on v_mfma_f32_32x32x2_f32
#   "f16x3" split precision: x = hi + lo (two fp16), Ahi.Bhi + Ahi.Blo + Alo.Bhi, fp32 accumulate
#   "f16"   the reference's autocast arithmetic (its AMP configs: trainer_template.py:211,281): operands rounded to fp16, one MFMA per
#           product, fp32 accumulate and epilogue; chained layers hand fp16 NDHWC tensors to each other (`out_split=True` then means
#           "fp16 output").  Inference only -- training-mode modules run the f16x3 kernels under this setting.
PRECISIONS = ("f32", "f16x3", "f16")
_precision = os.environ.get("OSA_PRECISION", "f32")
assert _precision in PRECISIONS, f"OSA_PRECISION must be one of {PRECISIONS}"


def set_precision(p: str):
    """Default mode for PackedConv3d objects created afterwards (call model.reset_engine() to repack)."""
    global _precision
    assert p in PRECISIONS, p
    _precision = p


def get_precision() -> str:
    return _precision


# Inside a `torch.autocast("cuda", dtype=torch.float16)` region the reference's convolutions multiply fp16 operands (its AMP configs:
# trainer_template.py:211,281).  With AUTOCAST_NATIVE the engine layers built / fetched in such a region use the "f16" mode -- the same
# arithmetic, one MFMA per product -- instead of the global mode (f16x3 would be correct too: 3x the matrix work for accuracy autocast
# has already given up).  bf16 autocast regions keep the global mode (no native bf16 kernels).  OSA_AUTOCAST_NATIVE=0 switches it off.
AUTOCAST_NATIVE = os.environ.get("OSA_AUTOCAST_NATIVE", "1") != "0"


def effective_precision() -> str:
    """arithmetic mode of engine layers packed NOW: the global mode, or "f16" inside an fp16 autocast region (AUTOCAST_NATIVE)"""
    if AUTOCAST_NATIVE and _precision != "f16" and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16 \
            and not torch.is_grad_enabled():
        return "f16"
    return _precision


AMP_TRAIN_NATIVE = os.environ.get("OSA_AMP_TRAIN_NATIVE", "1") != "0"


def train_precision(requested=None) -> str:
    """arithmetic mode of the DIFFERENTIABLE engine convolutions (autograd.py) called now: the requested / global mode, or "f16" inside an
    fp16 autocast region (AMP_TRAIN_NATIVE) -- the reference trains StereoBase / LightStereo / IGEV under autocast + GradScaler
    (trainer_template.py:211,217-226; cfgs/stereobase/stereobase_sceneflow.yaml:50), where every convolution, its data gradient and its
    weight gradient multiply fp16 operands and accumulate in fp32.  bf16 regions keep the global mode (no native bf16 kernels: the
    fp32-class modes are strictly more accurate).  A global "f16" mode trains natively as well."""
    p = requested or _precision
    if p == "f16":
        return "f16" if AMP_TRAIN_NATIVE else "f16x3"
    # the autocast override applies to the GLOBAL mode only: an explicit per-call precision (an exact-f32 layer or test inside an autocast
    # region) is honoured
    if requested is None and AMP_TRAIN_NATIVE and AUTOCAST_NATIVE and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16:
        return "f16"
    return p


# ----------------------------------------------------------------------------- packed-weight caches
class _PackEntry:
    __slots__ = ("slots", "modes", "key", "value", "event", "stream", "synced")


# Set by parallel.SubBatchStreams(n > 1): several streams of one process use the engine concurrently, so a packed form built on one
# stream must be ordered before its first use on every other stream (see cached_pack).  Single-stream processes skip the bookkeeping.
MULTI_STREAM = False


def _tensor_slots(mods):
    """(dict, name) of every parameter and buffer under `mods`: read through the owning module's dict on every
    check, so a replaced Parameter object is seen as well as an in-place update."""
    out = []
    for top in mods:
        for m in top.modules():
            out += [(m._parameters, n) for n, p in m._parameters.items() if p is not None]
            out += [(m._buffers, n) for n, b in m._buffers.items() if b is not None]
    return out


def cached_pack(owner, attr, build, mods=None):
    """Packed engine form of `owner`'s layers, rebuilt whenever a source tensor changed.

    Packing folds eval-mode BatchNorm statistics and re-orders weights once; the reference's trainer alternates
    train / eval every epoch, optimisers and load_state_dict update parameters in place and .to() moves them, so
    the cache key is (arithmetic mode, (data_ptr, version counter) of every parameter and buffer the packed form was
    built from).  ~0.15 us per tensor per forward; hipGraph replays never get here.  `owner.<attr> = None` (what the
    reset_engine() methods do) still forces a rebuild, e.g. after replacing a sub-module object.

    Streams (MULTI_STREAM): the pack kernels run on the stream that was current at build time.  A hit from ANOTHER stream first makes
    that stream wait for an event recorded behind the build (once per stream and build) -- without it a sub-batch stream could read
    packs that the first sub-batch's stream is still writing.  Inside a stream capture the wait is recorded only when the
    build itself was captured (fork / join edge of the graph); builds from an eager warm-up are complete by the time a capture starts
    (torch.cuda.graph synchronises first)."""
    ent = owner.__dict__.get(attr)
    if not isinstance(ent, _PackEntry):
        ent = _PackEntry()
        ent.slots, ent.modes = _tensor_slots(mods if mods is not None else (owner,)), {}
        object.__setattr__(owner, attr, ent)
    mode = effective_precision()
    key = []
    for d, n in ent.slots:
        t = d.get(n)
        if t is None:
            key.append(None)
        else:
            key.append(t.data_ptr()); key.append(t._version)
    # one slot PER arithmetic mode: a module called alternately inside and outside an fp16 autocast region (or with and
    # without grad) keeps both packed forms instead of rebuilding on every call -- and a rebuild in one mode never frees buffers that
    # another sub-batch stream, or a captured graph, of the other mode still reads.
    slot = ent.modes.get(mode)
    if slot is None:
        slot = ent.modes[mode] = _PackEntry()
        slot.key = slot.value = slot.event = slot.stream = slot.synced = None
    if slot.key != key:
        slot.value = build()
        slot.key = key
        slot.event = None
        if MULTI_STREAM and torch.cuda.is_available():
            cur = torch.cuda.current_stream()
            slot.event = torch.cuda.Event()
            slot.event.record(cur)
            slot.stream, slot.synced = cur.cuda_stream, (torch.cuda.is_current_stream_capturing(), set())
    elif slot.event is not None:
        cur = torch.cuda.current_stream()
        h = cur.cuda_stream
        if h != slot.stream and h not in slot.synced[1]:
            if slot.synced[0] or not torch.cuda.is_current_stream_capturing():
                cur.wait_event(slot.event)
            slot.synced[1].add(h)
    return slot.value


_EMPTY = {}


def _empty(device):
    """an empty fp32 tensor on `device` (stands for a NULL pointer in the extension's tensor lists)"""
    t = _EMPTY.get(device)
    if t is None:
        t = _EMPTY[device] = torch.empty(0, device=device, dtype=torch.float32)
    return t


_BATCHNORM = nn.modules.batchnorm._BatchNorm          # BatchNorm1d / 2d / 3d AND nn.SyncBatchNorm (what convert_sync_batchnorm leaves)
_INSTANCENORM = nn.modules.instancenorm._InstanceNorm


def norm_kind(n):
    """Classify the normalisation module behind a convolution at a pack site: None (no norm / nn.Identity), "bn" (any
    `_BatchNorm` subclass -- the reference's trainer converts every BatchNorm to nn.SyncBatchNorm before DDP when SYNC_BN is set,
    trainer_template.py:83-85, and SyncBatchNorm is NOT a BatchNorm2d / 3d -- folded from its running statistics) or "in"
    (InstanceNorm, a separate kernel).  Anything else RAISES: a pack site that cannot fold a norm must never drop it silently
    (VERDICT r4, weak #1 / #14)."""
    if n is None or isinstance(n, nn.Identity):
        return None
    if isinstance(n, _BATCHNORM):
        if n.running_mean is None or n.running_var is None:
            raise _lib.EngineError(f"{type(n).__name__}(track_running_stats=False) has no statistics to fold into the engine's conv launch")
        return "bn"
    if isinstance(n, _INSTANCENORM):
        return "in"
    raise _lib.EngineError(f"the engine cannot fold a {type(n).__name__} behind a convolution (BatchNorm / SyncBatchNorm / InstanceNorm only)")


def foldable_bn(n):
    """`n` if it is a BatchNorm the conv launch can fold (incl. SyncBatchNorm), None when there is no norm; raises otherwise."""
    k = norm_kind(n)
    if k == "in":
        raise _lib.EngineError("an InstanceNorm at a pack site that folds BatchNorm only")
    return n if k == "bn" else None


def bn_scale_shift(bn):
    """Eval-mode BatchNorm as y = x*scale + shift (eps from the module, default 1e-5)."""
    bn = foldable_bn(bn)
    if bn is None:
        return None, None
    if bn.training:
        raise _lib.EngineError(f"{type(bn).__name__} in training mode reached an eval-mode pack site: batch statistics cannot be folded "
                               "(call .eval() on the module, or FREEZE_BN, or take the training path)")
    w = bn.weight if bn.weight is not None else torch.ones_like(bn.running_mean)
    b = bn.bias if bn.bias is not None else torch.zeros_like(bn.running_mean)
    scale = (w.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)).contiguous()
    shift = (b.detach().float() - bn.running_mean.detach().float() * scale).contiguous()
    return scale, shift


def _t3(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


class PackedConv3d:
    """conv (or stride-2 transposed conv) + folded BN + activation, weights in MFMA operand order."""

    def __init__(self, conv, bn=None, act=ACT_NONE, slope=0.01, precision=None):
        self.precision = precision or effective_precision()
        assert self.precision in PRECISIONS
        w = conv.weight.detach()
        if not w.is_cuda:
            raise _lib.EngineError("PackedConv3d needs parameters on the GPU (no CPU path)")
        w = w.float().contiguous()
        self.transposed = isinstance(conv, (nn.ConvTranspose3d, nn.ConvTranspose2d))
        self.flat_deconv = isinstance(conv, nn.ConvTranspose2d)    # D = 1 map, 4 parity classes
        if isinstance(conv, (nn.Conv2d, nn.ConvTranspose2d)):            # 2-D layer == 3-D layer with D = 1 and a 1 x kh x kw kernel
            w = w[:, :, None].contiguous()
            self.k = (1,) + tuple(conv.kernel_size)
            self.stride = (1,) + tuple(conv.stride)
            self.pad = (0,) + tuple(conv.padding)
            self.dil = (1,) + tuple(conv.dilation)
        else:
            self.k = tuple(conv.kernel_size)
            self.stride = _t3(conv.stride)
            self.pad = _t3(conv.padding)
            self.dil = _t3(conv.dilation)
        self.act, self.slope = act, float(slope)
        self.scale, self.shift = bn_scale_shift(bn)
        if conv.bias is not None:
            bias = conv.bias.detach().float()
            if bn is None:
                self.shift, self.scale = bias.contiguous(), torch.ones_like(bias)
            else:                                   # BN(conv + bias) = conv*s + (bias*s + t)
                self.shift = (self.shift + bias * self.scale).contiguous()
        # output bound of this layer for the f16x3 range tracking: max_co |bn scale| * sum|w_co|, max_co |bn shift|
        wsum = w.abs().sum(dim=(0, 2, 3, 4)) if isinstance(conv, (nn.ConvTranspose3d, nn.ConvTranspose2d)) else w.abs().sum(dim=(1, 2, 3, 4))
        gain = (wsum * self.scale.abs()).amax() if self.scale is not None else wsum.amax()
        smax = self.shift.abs().amax() if self.shift is not None else torch.zeros((), device=w.device)
        self.coef = torch.stack([gain.float(), smax.float()]).contiguous()
        st = _stream()
        f16 = self.precision == "f16x3"
        h16 = self.precision == "f16"
        self.out_scale = 1.0
        wscale = 1.0
        if f16:
            # power-of-two pre-scale: largest |w| lands in [2^13, 2^14) -> hi AND lo parts are fp16 normals
            amax = float(w.abs().max())
            k = 0 if amax == 0.0 or not math.isfinite(amax) else int(math.floor(math.log2(16384.0 / amax)))
            wscale = 2.0 ** max(-14, min(k, 40))
            self.out_scale = 1.0 / wscale
        if self.transposed:
            self.Ci, self.Co = w.shape[0], w.shape[1]
            assert conv.groups == 1
            if self.flat_deconv:
                assert self.k[1] == self.k[2] and self.stride == (1, 2, 2)
                self.opad = (0,) + tuple(conv.output_padding)
                fam = "osa_deconv2d"
            else:
                assert self.k[0] == self.k[1] == self.k[2] and self.stride == (2, 2, 2)
                self.opad = _t3(conv.output_padding)
                fam = "osa_deconv3d"
            n = getattr(_lib.load(), fam + "_packed_floats")(self.Ci, self.Co, self.k[1])
            self.packed = torch.zeros(n, device=w.device, dtype=torch.float32)
            ext = _ext.load()
            if ext is not None:
                ext.weight_pack(w, self.packed, 2 if self.flat_deconv else 1, PRECISIONS.index(self.precision), [self.Ci, self.Co, self.k[1], self.pad[1]], wscale)
            elif f16:
                _lib.call(fam + "_pack_f16x3", w.data_ptr(), self.packed.data_ptr(), self.Ci, self.Co,
                          self.k[1], self.pad[1], wscale, st)
            elif h16:
                _lib.call(fam + "_pack_f16", w.data_ptr(), self.packed.data_ptr(), self.Ci, self.Co, self.k[1], self.pad[1], st)
            else:
                _lib.call(fam + "_pack_f32", w.data_ptr(), self.packed.data_ptr(), self.Ci, self.Co,
                          self.k[1], self.pad[1], st)
        else:
            self.Co, self.Ci = w.shape[0], w.shape[1]
            assert conv.groups == 1
            s = self.stride
            assert s[1] == s[2] and (s[0] == s[1] or (self.k[0] == 1)), f"anisotropic stride {s}"
            n = _lib.load().osa_conv3d_packed_floats(self.Ci, self.Co, *self.k)
            self.packed = torch.zeros(n, device=w.device, dtype=torch.float32)
            ext = _ext.load()
            if ext is not None:
                ext.weight_pack(w, self.packed, 0, PRECISIONS.index(self.precision), [self.Ci, self.Co, *self.k], wscale)
            elif f16:
                _lib.call("osa_conv3d_pack_f16x3", w.data_ptr(), self.packed.data_ptr(), self.Ci, self.Co, *self.k, wscale, st)
            elif h16:
                _lib.call("osa_conv3d_pack_f16", w.data_ptr(), self.packed.data_ptr(), self.Ci, self.Co, *self.k, st)
            else:
                _lib.call("osa_conv3d_pack_f32", w.data_ptr(), self.packed.data_ptr(), self.Ci, self.Co, *self.k, st)

    def out_shape(self, D, H, W):
        if self.transposed:
            k, p, op = self.k[1], self.pad[1], self.opad[1]
            up = lambda n: (n - 1) * 2 - 2 * p + k + op
            return (1, up(H), up(W)) if self.flat_deconv else (up(D), up(H), up(W))
        s = self.stride[1]
        sd = 1 if (D == 1 and self.k[0] == 1) else s
        f = lambda n, k, p, d, st: (n + 2 * p - d * (k - 1) - 1) // st + 1
        return (f(D, self.k[0], self.pad[0], self.dil[0], sd), f(H, self.k[1], self.pad[1], self.dil[1], s),
                f(W, self.k[2], self.pad[2], self.dil[2], s))

    def __call__(self, x, residual=None, out=None, gate=None, x_off=0, out_off=0, res_off=0, gate_raw=False, redir=None,
                 out_split=False, gate_channels=0, res_after_act=False):
        """x: logical [B,Cs>=Ci,D,H,W] NDHWC; channels [x_off, x_off+Ci) are read (x_off % 4 == 0).
        Returns logical [B,Co(pad 4),Do,Ho,Wo] NDHWC, or writes channels [out_off, out_off+Co) of `out`
        (channel-slice output replaces torch.cat).  gate: NHWC logits [B,Ho,Wo,>=Co]; the result is
        multiplied by sigmoid(gate) broadcast over D (FeatureAtt); gate_raw=True multiplies by the
        gate itself (LightStereo AttentionModule: attn * cost); gate_channels=n gates output channels [0, n) only
        (OSA_GATE_CHANNELS: the fused ConvGRU r|z launch).  redir=(layer, t): a transposed conv adds
        layer(t) -- a 1x1x1 PackedConv3d (+BN) on the output-resolution tensor t (<= 64 channels) -- inside
        its epilogue (GwcNet hourglass conv6 + redir1); replaces `residual`.  out_split=True (f16x3 only)
        writes the output as a split tensor (see IN_SPLIT ...); split inputs / residuals are recognised by
        their `_osa_split` tag, so chains of engine layers pass them along without further arguments."""
        h16 = self.precision == "f16"
        assert is_cl(x) and (x.dtype == torch.float32 or (h16 and x.dtype == torch.float16)), "engine tensors are fp32 NDHWC (f16 mode: fp32 or fp16)"
        B, Cs, D, H, W = x.shape
        cpad = 8 if x.dtype == torch.float16 else 4         # 16-byte channel rows
        assert Cs >= x_off + self.Ci and Cs % cpad == 0 and x_off % cpad == 0, f"input has {Cs} channels, layer expects {self.Ci}"
        Ci = min((self.Ci + cpad - 1) // cpad * cpad, Cs - x_off)     # padded channels of x are zero by construction
        Do, Ho, Wo = self.out_shape(D, H, W)
        if out is None:
            # f16 mode: fp16 output where the chain asks for it and the kernel can write it (complete 8-channel rows, fp16 residual, no gate)
            o16 = h16 and out_split and self.Co % 8 == 0 and gate is None and (residual is None or residual.dtype == torch.float16)
            CoS = (self.Co + 7) // 8 * 8 if o16 else (self.Co + 3) // 4 * 4
            out = empty_cl(B, CoS, Do, Ho, Wo, x.device, torch.float16 if o16 else torch.float32)
            if CoS != self.Co:
                out.zero_()
        if h16:
            out_split = out.dtype == torch.float16          # f16 mode: the chain format is the dtype of the output buffer
        assert is_cl(out) and tuple(out.shape[2:]) == (Do, Ho, Wo) and out.shape[1] >= out_off + self.Co
        yCs = out.shape[1]
        rCs = 0
        if residual is not None:
            assert is_cl(residual) and tuple(residual.shape[2:]) == (Do, Ho, Wo)
            rCs = residual.shape[1]
            assert rCs >= res_off + self.Co
            assert h16 or residual.dtype == torch.float32, "fp16 chain tensors exist in the f16 mode only (the launch would read them as fp32)"
        gCs = 0
        if gate is not None:
            assert gate.is_contiguous() and tuple(gate.shape[:3]) == (B, Ho, Wo) and gate.shape[3] >= self.Co
            assert h16 or gate.dtype == torch.float32
            gCs = gate.shape[3]
        ext = _ext.load()
        xp = yp = rp = None                                  # raw addresses: the ctypes path only (FakeTensors have none: tests/test_gpu_fake_trace.py)
        if ext is None:
            xp, yp = x.data_ptr() + x.element_size() * x_off, out.data_ptr() + out.element_size() * out_off
            rp = None if residual is None else residual.data_ptr() + residual.element_size() * res_off
        act = self.act | (GATE_RAW if (gate is not None and gate_raw) else 0)
        if gate_channels:
            assert gate is not None and gate_channels % 4 == 0 and 0 < gate_channels <= self.Co
            act |= gate_channels << 16
        if res_after_act:
            assert residual is not None and self.act == ACT_RELU and gate is None and not out_split and not self.transposed
            act |= RES_AFTER_ACT
        fmt = (IN_SPLIT if is_split(x) else 0) | (OUT_SPLIT if out_split else 0) | (RES_SPLIT if is_split(residual) else 0) \
            | (REDIR_SPLIT if (redir is not None and is_split(redir[1])) else 0)
        if fmt and h16:
            assert redir is None and (gate is None or not out_split)
            assert not (fmt & RES_SPLIT) or res_off % 4 == 0
            assert not out_split or (self.Co % 8 == 0 and yCs % 8 == 0 and out_off % 8 == 0 and (residual is None or residual.dtype == torch.float16))
            act |= fmt
        elif fmt:
            assert self.precision == "f16x3", "split activation tensors exist in the f16x3 mode only"
            assert gate is None
            assert not (fmt & IN_SPLIT) or x_off % 16 == 0          # split layout is per 16-channel chunk
            assert not (fmt & RES_SPLIT) or res_off % 16 == 0
            assert not out_split or (self.Co % 16 == 0 and yCs % 16 == 0 and out_off % 16 == 0)
            act |= fmt
        rng, st = None, _stream()
        if self.precision == "f16x3":
            # operand ranges (device-side): scale of x / residual / redir input, bound for a split output, and the
            # output's own running maximum
            need_res = residual is not None and (out_split or is_split(residual))    # its scale (split) / its share of the output bound
            mx, mr, mo = input_meta(x), (input_meta(residual) if need_res else None), attach_meta(out, st)
            mrd = None if redir is None else input_meta(redir[1])
            if ext is None:
                rng = _lib.F16x3Ranges(mx.data_ptr(), None if mr is None else mr.data_ptr(),
                                       None if mrd is None else mrd.data_ptr(), mo.data_ptr(),
                                       self.coef.data_ptr(), None if redir is None else redir[0].coef.data_ptr())
        taps = self.k[0] * self.k[1] * self.k[2]
        macs = B * Do * Ho * Wo * self.Ci * self.Co * taps / ((4 if self.flat_deconv else 8) if self.transposed else 1)
        nbytes = 4 * B * (D * H * W * self.Ci + Do * Ho * Wo * self.Co * (1 + (residual is not None) + (redir is not None)))
        with timing.span("deconv3d" if self.transposed else "conv3d", self.Ci, self.Co, self.k[0], self.stride[1], D, H, W,
                         flops=2 * macs, nbytes=nbytes):
            tail = (self.out_scale, rng, st) if self.precision == "f16x3" else (st,)
            sfx = self.precision
            if ext is not None and redir is not None:
                assert not h16, "the f16 mode has no fused redir branch (run the 1x1x1 layer and pass it as residual)"
                rl, rt = redir
                assert self.transposed and not self.flat_deconv and residual is None and gate is None
                assert rl.precision == self.precision and rl.k == (1, 1, 1) and rl.Co == self.Co and rl.act == ACT_NONE
                assert is_cl(rt) and tuple(rt.shape[2:]) == (Do, Ho, Wo) and rt.shape[1] >= rl.Ci and rl.Ci <= 64
                if self.precision == "f16x3":
                    e = _empty(x.device)
                    metas = [mx, e, mrd, mo, self.coef, rl.coef, e]
                else:
                    metas = []
                ext.deconv_redir(x, x_off, self.packed, self.scale, self.shift, out, out_off, [B, D, H, W, Ci, Cs, self.Co, yCs],
                                 [self.k[0], self.pad[0], self.opad[0]], rt, [rt.shape[1], (rl.Ci + 3) // 4 * 4], rl.packed, rl.scale, rl.shift,
                                 rl.out_scale, PRECISIONS.index(self.precision), act, self.slope, self.out_scale, metas)
            elif ext is not None:
                # PyTorch-ROCm C++ extension (csrc/torch_ext.cpp): one dispatcher call, tensors in, current HIP stream inside
                if self.transposed:
                    fam, geom = (2 if self.flat_deconv else 1), [self.k[1], self.pad[1], self.opad[1]]
                else:
                    fam, geom = 0, [self.k[0], self.k[1], self.k[2], self.stride[1], self.pad[0], self.pad[1], self.pad[2], self.dil[0], self.dil[1], self.dil[2]]
                if self.precision == "f16x3":
                    e = _empty(x.device)
                    metas = [mx, e if mr is None else mr, e, mo, self.coef, e, e]
                else:
                    metas = []
                ext.conv_ndhwc(x, x_off, self.packed, self.scale, self.shift, residual, res_off, out, out_off, gate,
                               [B, D, H, W, Ci, Cs, self.Co, yCs, rCs, gCs], geom, fam, PRECISIONS.index(self.precision), act, self.slope,
                               self.out_scale, metas)
            elif redir is not None:
                assert not h16, "the f16 mode has no fused redir branch (run the 1x1x1 layer and pass it as residual)"
                rl, rt = redir
                assert self.transposed and not self.flat_deconv and residual is None and gate is None
                assert rl.precision == self.precision and rl.k == (1, 1, 1) and rl.Co == self.Co and rl.act == ACT_NONE
                assert is_cl(rt) and tuple(rt.shape[2:]) == (Do, Ho, Wo) and rt.shape[1] >= rl.Ci and rl.Ci <= 64
                rtail = (rl.out_scale,) if self.precision == "f16x3" else ()
                _lib.call("osa_deconv3d_redir_ndhwc_" + sfx, xp, self.packed.data_ptr(), _p(self.scale), _p(self.shift),
                          yp, B, D, H, W, Ci, Cs, self.Co, yCs, self.k[0], self.pad[0], self.opad[0],
                          rt.data_ptr(), rt.shape[1], (rl.Ci + 3) // 4 * 4, rl.packed.data_ptr(), _p(rl.scale), _p(rl.shift), *rtail,
                          act, self.slope, *tail)
            elif self.flat_deconv:
                assert D == 1
                _lib.call("osa_deconv2d_nhwc_" + sfx, xp, self.packed.data_ptr(), _p(self.scale), _p(self.shift),
                          rp, yp, B, H, W, Ci, Cs, self.Co, yCs, rCs,
                          self.k[1], self.pad[1], self.opad[1], _p(gate), gCs, act, self.slope, *tail)
            elif self.transposed:
                _lib.call("osa_deconv3d_ndhwc_" + sfx, xp, self.packed.data_ptr(), _p(self.scale), _p(self.shift),
                          rp, yp, B, D, H, W, Ci, Cs, self.Co, yCs, rCs,
                          self.k[0], self.pad[0], self.opad[0], _p(gate), gCs, act, self.slope, *tail)
            else:
                _lib.call("osa_conv3d_ndhwc_" + sfx, xp, self.packed.data_ptr(), _p(self.scale), _p(self.shift),
                          rp, yp, B, D, H, W, Ci, Cs, self.Co, yCs, rCs,
                          self.k[0], self.k[1], self.k[2], self.stride[1],
                          self.pad[0], self.pad[1], self.pad[2], self.dil[0], self.dil[1], self.dil[2],
                          _p(gate), gCs, act, self.slope, *tail)
        if out_split and not h16:
            out._osa_split = True
        return out


class DepthwiseConv2d:
    """Depthwise nn.Conv2d (groups == channels) + folded BN / bias + activation on an NHWC map
    (logical [B,C,1,H,W] NDHWC tensor).  LightStereo MobileV2Residual.dwconv and the AttentionModule
    strip convolutions (aggregation.py:79-83, 105-113).  fp32 fmaf per tap in both precision modes."""

    def __init__(self, conv, bn=None, act=ACT_NONE):
        assert isinstance(conv, nn.Conv2d) and conv.groups == conv.in_channels == conv.out_channels
        w = conv.weight.detach()
        if not w.is_cuda:
            raise _lib.EngineError("DepthwiseConv2d needs parameters on the GPU (no CPU path)")
        w = w.float().contiguous()
        self.C = conv.in_channels
        assert self.C % 4 == 0, "depthwise engine layers need a channel count divisible by 4"
        self.k, self.stride, self.pad, self.dil = tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding), tuple(conv.dilation)
        assert self.stride[0] == self.stride[1]
        self.act = act
        self.scale, self.shift = bn_scale_shift(bn)
        if conv.bias is not None:
            bias = conv.bias.detach().float()
            if bn is None:
                self.shift, self.scale = bias.contiguous(), None
            else:
                self.shift = (self.shift + bias * self.scale).contiguous()
        self.packed = torch.empty(self.k[0] * self.k[1] * self.C, device=w.device, dtype=torch.float32)
        ext = _ext.load()
        if ext is not None:
            ext.weight_pack(w, self.packed, 3, 0, [self.C, self.k[0], self.k[1]], 1.0)
        else:
            _lib.call("osa_dwconv2d_pack_f32", w.data_ptr(), self.packed.data_ptr(), self.C, self.k[0], self.k[1], _stream())

    def __call__(self, x, add=None, out_f16=False):
        """out_f16 / an fp16 `x` (r6, the 3 x 3 layers in the f16 mode): the chain tensors between MobileV2Residual's expansion, depthwise and
        projection convolutions are fp16 -- what the reference's autocast moves there -- instead of fp32 (half the bytes of HBM-bound layers)"""
        assert is_cl(x) and x.dtype in (torch.float32, torch.float16) and x.shape[2] == 1
        B, Cs, _, H, W = x.shape
        assert Cs >= self.C
        f = lambda n, k, p, d, st: (n + 2 * p - d * (k - 1) - 1) // st + 1
        Ho, Wo = f(H, self.k[0], self.pad[0], self.dil[0], self.stride[0]), f(W, self.k[1], self.pad[1], self.dil[1], self.stride[1])
        h16 = x.dtype == torch.float16 or out_f16
        if h16:
            assert add is None and self.k[1] == 3 and self.dil == (1, 1), "fp16 tensors: the 3 x 3 depthwise layers"
        out = empty_cl(B, self.C, 1, Ho, Wo, x.device, torch.float16 if out_f16 else torch.float32)
        aCs = 0
        if add is not None:
            assert is_cl(add) and tuple(add.shape[2:]) == (1, Ho, Wo) and add.shape[1] >= self.C
            aCs = add.shape[1]
        with timing.span("dwconv2d", self.C, self.C, self.k[0] * self.k[1], self.stride[0], 1, H, W):
            ext = _ext.load()
            if h16:
                if ext is not None:
                    ext.dwconv2d_f16io(x, self.packed, self.scale, self.shift, out, [B, H, W, self.C, Cs, self.C],
                                       [self.k[0], self.k[1], self.stride[0], self.pad[0], self.pad[1]], self.act, attach_meta(out))
                else:
                    _lib.call("osa_dwconv2d_nhwc_f16io", x.data_ptr(), int(x.dtype == torch.float16), self.packed.data_ptr(), _p(self.scale), _p(self.shift),
                              out.data_ptr(), int(out_f16), B, H, W, self.C, Cs, self.C, self.k[0], self.k[1], self.stride[0], self.pad[0], self.pad[1],
                              self.act, attach_meta(out).data_ptr(), _stream())
            elif ext is not None:
                ext.dwconv2d(x, self.packed, self.scale, self.shift, add, out, [B, H, W, self.C, Cs, self.C, aCs],
                             [self.k[0], self.k[1], self.stride[0], self.pad[0], self.pad[1], self.dil[0], self.dil[1]], self.act, attach_meta(out))
            else:
                _lib.call("osa_dwconv2d_nhwc_f32", x.data_ptr(), self.packed.data_ptr(), _p(self.scale), _p(self.shift),
                          _p(add), out.data_ptr(), B, H, W, self.C, Cs, self.C, aCs,
                          self.k[0], self.k[1], self.stride[0], self.pad[0], self.pad[1], self.dil[0], self.dil[1],
                          self.act, attach_meta(out).data_ptr(), _stream())
        return out


class SmallCoConv3d:
    """'same' convolution with <= 4 output channels (the 32->1 classifier heads); reads the
    reference-layout weight directly."""

    def __init__(self, conv):
        assert isinstance(conv, nn.Conv3d) and conv.out_channels <= 4
        assert _t3(conv.stride) == (1, 1, 1) and _t3(conv.dilation) == (1, 1, 1)
        self.w = conv.weight.detach().float().contiguous()
        if not self.w.is_cuda:
            raise _lib.EngineError("SmallCoConv3d needs parameters on the GPU (no CPU path)")
        self.bias = None if conv.bias is None else conv.bias.detach().float().contiguous()
        self.Co, self.Ci = self.w.shape[:2]
        self.k, self.pad = tuple(conv.kernel_size), _t3(conv.padding)
        n = _lib.load().osa_conv3d_small_co_packed_floats(self.Ci, self.Co, *self.k)
        self.packed = torch.empty(n + 16, device=self.w.device, dtype=torch.float32)     # scalar-cache friendly layout
        off = (-self.packed.data_ptr() // 4) % 16                                         # 64-byte alignment
        self.packed = self.packed[off:off + n]
        ext = _ext.load()
        if ext is not None:
            ext.weight_pack(self.w, self.packed, 4, 0, [self.Ci, self.Co, *self.k], 1.0)
        else:
            _lib.call("osa_conv3d_small_co_pack_f32", self.w.data_ptr(), self.packed.data_ptr(), self.Ci, self.Co, *self.k, _stream())

    def __call__(self, x, residual=None):
        """x NDHWC logical [B,Cs,D,H,W] -> logical [B,Co,D,H,W] stored [B,D,H,W,Co] (for Co==1 this
        is plain contiguous [B,1,D,H,W]).  residual: an earlier output of the same shape (added)."""
        assert is_cl(x) and not is_split(x), "SmallCoConv3d reads fp32 NDHWC tensors"
        B, Cs, D, H, W = x.shape
        y = torch.empty((B, D, H, W, self.Co), device=x.device, dtype=torch.float32)
        if residual is not None:
            assert tuple(residual.shape) == (B, self.Co, D, H, W) and is_cl(residual)
        with timing.span("conv3d_small_co", self.Ci, self.Co, self.k[0], 1, D, H, W):
            ext = _ext.load()
            if ext is not None:
                ext.small_co_conv(x, self.packed, self.bias, residual, y, [B, D, H, W, self.Ci, Cs, self.Co, self.Co], [*self.k, *self.pad])
            else:
                _lib.call("osa_conv3d_small_co_packed_ndhwc_f32", x.data_ptr(), self.packed.data_ptr(), _p(self.bias), _p(residual), y.data_ptr(),
                          B, D, H, W, self.Ci, Cs, self.Co, self.Co, *self.k, *self.pad, _stream())
        return y.permute(0, 4, 1, 2, 3)
