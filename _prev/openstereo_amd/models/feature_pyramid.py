"""The 2-D feature pyramids of BASELINE configs[2]-[4]: `Feature` of StereoBase (models/stereobase/backbone.py:32-73) and IGEV-Stereo
(models/igev/extractor.py:320-355), `Backbone` of LightStereo (models/lightstereo/backbone.py:29-75).

Each is a timm MobileNetV2-100 trunk (`conv_stem`, `bn1`, `act1`, `model.blocks` regrouped into `block0..block4`) plus an FPN decoder that
IS reference code: Conv2xUp / Conv2x_IN / FPNLayer + a final 3x3 conv, InstanceNorm2d or BatchNorm2d, LeakyReLU.

* Decoders: mirrored with the reference's attribute names (identical `state_dict` keys) and pinned against the reference's own classes
  built around the same trunk (tests/golden/feature_pyramid.npz, make_golden.gen_feature_pyramid: `timm.create_model` answered by the
  mirror below).
* Trunk: `timm` and its pretrained weights are not available offline, so `MobileNetV2Trunk` restates the architecture timm builds for
  'mobilenetv2_100' (features_only) under timm's parameter names -- **parity unpinned -- timm absent**: layer list and key names follow
  timm 0.9's `efficientnet_builder` decode of ['ds_r1_k3_s1_c16', 'ir_r2_k3_s2_e6_c24', 'ir_r3_k3_s2_e6_c32', 'ir_r4_k3_s2_e6_c64',
  'ir_r3_k3_s1_e6_c96', 'ir_r3_k3_s2_e6_c160', 'ir_r1_k3_s1_e6_c320'] (stem 32, ReLU6), which a real checkpoint's `feature.*` keys load
  into; it is what the `*_e2e` bench workloads run instead of the 5-conv stand-in of round 3.

Inference on a GPU runs the whole pyramid on the engine (`forward` -> `forward_cl`): NHWC maps, every conv + BatchNorm (+ ReLU6 /
LeakyReLU, + residual) one fused MFMA launch, depthwise 3x3 on `dwconv2d_nhwc_kernel`, InstanceNorm + LeakyReLU on csrc/norm.hip, the
decoder's `torch.cat` replaced by channel-slice outputs.  Training mode / CPU tensors keep the torch composition (the reference's own op
sequence)."""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _ext, _lib, amp
from ..engine import (ACT_LEAKY, ACT_NONE, ACT_RELU6, DepthwiseConv2d, PackedConv3d, cached_pack, norm_kind)
from ..ops import _stream, empty_cl, is_cl, on_engine
from ..ranges import attach_meta, combine_meta, input_meta, meta_of
from .igev_style import BasicConv2d
from .lightstereo import cl_to_nchw, nchw_to_cl


# ----------------------------------------------------------------------------- timm MobileNetV2-100 trunk (key-compatible mirror)
class DepthwiseSeparableConv(nn.Module):
    """timm `DepthwiseSeparableConv` (blocks.0.0): dw 3x3 + BN + ReLU6, pw 1x1 + BN (no activation: pw_act = False)."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv_dw = nn.Conv2d(cin, cin, 3, stride, 1, groups=cin, bias=False)
        self.bn1 = nn.BatchNorm2d(cin)
        self.conv_pw = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.has_skip = stride == 1 and cin == cout
        self._eng = None

    def forward(self, x):
        y = self.bn2(self.conv_pw(F.relu6(self.bn1(self.conv_dw(x)))))
        return x + y if self.has_skip else y

    def forward_cl(self, x):
        dw, pw = cached_pack(self, "_eng", lambda: (DepthwiseConv2d(self.conv_dw, self.bn1, ACT_RELU6), PackedConv3d(self.conv_pw, self.bn2, ACT_NONE)))
        return pw(dw(x), residual=x if self.has_skip else None)


class InvertedResidual(nn.Module):
    """timm `InvertedResidual`: pw expand + BN + ReLU6, dw 3x3 + BN + ReLU6, pw-linear + BN, skip when stride 1 and cin == cout."""

    def __init__(self, cin, cout, stride, expand=6):
        super().__init__()
        mid = cin * expand
        self.conv_pw = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv_dw = nn.Conv2d(mid, mid, 3, stride, 1, groups=mid, bias=False)
        self.bn2 = nn.BatchNorm2d(mid)
        self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.has_skip = stride == 1 and cin == cout
        self._eng = None

    def forward(self, x):
        y = F.relu6(self.bn1(self.conv_pw(x)))
        y = F.relu6(self.bn2(self.conv_dw(y)))
        y = self.bn3(self.conv_pwl(y))
        return x + y if self.has_skip else y

    def forward_cl(self, x):
        pw, dw, pl = cached_pack(self, "_eng", lambda: (PackedConv3d(self.conv_pw, self.bn1, ACT_RELU6), DepthwiseConv2d(self.conv_dw, self.bn2, ACT_RELU6),
                                                        PackedConv3d(self.conv_pwl, self.bn3, ACT_NONE)))
        return pl(dw(pw(x)), residual=x if self.has_skip else None)


_MBV2_ARCH = ((16, 1, 1, 1), (24, 2, 2, 6), (32, 3, 2, 6), (64, 4, 2, 6), (96, 3, 1, 6), (160, 3, 2, 6), (320, 1, 1, 6))   # (channels, repeats, stride, expand)


class MobileNetV2Trunk(nn.Module):
    """What `timm.create_model('mobilenetv2_100', features_only=True)` hands the reference: `.conv_stem`, `.bn1`, `.act1`, `.blocks`
    (7 stages).  Parity unpinned -- timm absent (module docstring)."""

    def __init__(self):
        super().__init__()
        self.conv_stem = nn.Conv2d(3, 32, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(32)
        self.act1 = nn.ReLU6(inplace=True)
        stages, cin = [], 32
        for i, (c, r, s, e) in enumerate(_MBV2_ARCH):
            blocks = []
            for j in range(r):
                blocks.append(DepthwiseSeparableConv(cin, c, s if j == 0 else 1) if i == 0 else InvertedResidual(cin, c, s if j == 0 else 1, e))
                cin = c
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)


def create_model(name="mobilenetv2_100", pretrained=False, features_only=True, **kw):
    """Stand-in for `timm.create_model` (the fixtures' fake `timm` module points here): only the trunk the three pyramids ask for."""
    if name != "mobilenetv2_100":
        raise NotImplementedError(f"feature_pyramid.create_model: '{name}' (only mobilenetv2_100 is mirrored; timm is not available offline)")
    return MobileNetV2Trunk()


# ----------------------------------------------------------------------------- engine helpers
def instance_norm_act_cl(x, C, act=ACT_LEAKY, slope=0.01, out=None, out_off=0, eps=1e-5):
    """InstanceNorm2d(affine=False) + activation of channels [0, C) of the NHWC map x (logical [B,Cs,1,H,W]); writes channels
    [out_off, out_off + C) of `out` (a concat buffer) or a fresh map.  csrc/norm.hip, deterministic."""
    assert is_cl(x) and x.dtype == torch.float32 and x.shape[2] == 1
    B, Cs, _, H, W = x.shape
    if out is None:
        out = empty_cl(B, (C + 3) // 4 * 4, 1, H, W, x.device)
    assert is_cl(out) and tuple(out.shape[2:]) == (1, H, W) and out.shape[1] >= out_off + C and out_off % 4 == 0
    lib = _lib.load()
    ws = torch.empty(lib.osa_instnorm_workspace_floats(B, H * W, C), device=x.device, dtype=torch.float32)
    m = meta_of(out)
    if m is None and meta_of(x) is not None:
        m = attach_meta(out)
    ext = _ext.load()
    if ext is not None:
        ext.instnorm_nhwc(x, out, out_off, [B, H * W, C, Cs, out.shape[1]], float(eps), act, float(slope), ws, m)
    else:
        _lib.call("osa_instnorm_nhwc_f32", x.data_ptr(), out.data_ptr() + 4 * out_off, B, H * W, C, Cs, out.shape[1], float(eps), act, float(slope),
                  ws.data_ptr(), None if m is None else m.data_ptr(), _stream())
    return out


class _ConvNormAct:
    """Engine form of one `conv (+ BatchNorm | InstanceNorm) (+ LeakyReLU)` unit of the decoders.  BatchNorm folds into the conv launch;
    InstanceNorm needs the whole map: conv launch, then csrc/norm.hip.  `replicate`: the conv pads by edge replication
    (lightstereo/backbone.py:57 `padding_mode="replicate"`): the map is padded first (one small torch op), the conv runs unpadded."""

    def __init__(self, conv, norm, act, slope):
        self.inorm = norm_kind(norm) == "in"
        self.replicate = getattr(conv, "padding_mode", "zeros") == "replicate"
        self.act, self.slope = act, slope
        c = conv
        if self.replicate:
            c = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, 0, conv.dilation, bias=conv.bias is not None).to(conv.weight.device)
            c.weight, c.bias = conv.weight, conv.bias
            self.pad = tuple(conv.padding)
        if self.inorm:
            self.conv = PackedConv3d(c, None, ACT_NONE)
        else:
            self.conv = PackedConv3d(c, norm, act, slope)
        self.Co = conv.out_channels

    def __call__(self, x, out=None, out_off=0):
        if self.replicate:
            B, Cs, _, H, W = x.shape
            xp = F.pad(x.permute(0, 2, 3, 4, 1).reshape(B, H, W, Cs).permute(0, 3, 1, 2), (self.pad[1], self.pad[1], self.pad[0], self.pad[0]), mode="replicate")
            x = xp.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).unsqueeze(2)
        if not self.inorm:
            return self.conv(x, out=out, out_off=out_off)
        return instance_norm_act_cl(self.conv(x), self.Co, self.act, self.slope, out=out, out_off=out_off)


def _unit(block, slope_default=0.01):
    """engine form of a reference unit: BasicConv2d / BasicDeconv2d (`.block` = [conv, norm?, act?]) or BasicConvIN (`.conv`, `.IN`)"""
    if hasattr(block, "block"):
        layers = list(block.block)
        conv = layers[0]
        actl = next((l for l in layers[1:] if isinstance(l, (nn.LeakyReLU, nn.ReLU))), None)
        norms = [l for l in layers[1:] if l is not actl]
        assert len(norms) <= 1, "BasicConv2d / BasicDeconv2d block = [conv, norm?, act?]"
        norm = norms[0] if norms else None      # classified by engine.norm_kind in _ConvNormAct: any _BatchNorm (SyncBatchNorm too) folds, unknown norms raise
        if actl is None:
            return _ConvNormAct(conv, norm, ACT_NONE, 0.0)
        if isinstance(actl, nn.LeakyReLU):
            return _ConvNormAct(conv, norm, ACT_LEAKY, actl.negative_slope)
        return _ConvNormAct(conv, norm, 1, 0.0)
    return _ConvNormAct(block.conv, block.IN if block.use_in else None, ACT_LEAKY if block.relu else ACT_NONE, slope_default)


def _up_cat(u1, u2, x, rem, order):
    """deconv unit u1 on x, concat with the skip `rem` ("xr": cat(x, rem) -- Conv2xUp / Conv2x_IN; "rx": cat(rem, x) -- FPNLayer) written
    as channel slices of one buffer, conv unit u2 on it"""
    B, _, _, H, W = rem.shape
    C1, Cr = u1.Co, rem.shape[1]
    buf = empty_cl(B, (C1 + Cr + 3) // 4 * 4, 1, H, W, rem.device)
    if (C1 + Cr) % 4:
        buf.zero_()
    o1, orr = (0, C1) if order == "xr" else (Cr, 0)
    assert o1 % 4 == 0 and orr % 4 == 0, "decoder channel counts are multiples of 4"
    y = u1(x, out=buf, out_off=o1)
    assert tuple(y.shape[2:]) == (1, H, W), "odd skip sizes (nearest resize in Conv2xUp.forward) take the torch path"
    buf[:, orr:orr + Cr] = rem[:, :Cr]
    if meta_of(buf) is not None:                      # f16x3 range block of the concatenation: the producer's maximum and the skip's
        buf._osa_meta = combine_meta(meta_of(buf), input_meta(rem))
    return u2(buf)


# ----------------------------------------------------------------------------- the three pyramids
class _Pyramid(nn.Module):
    def _take_trunk(self, model, groups):
        self.conv_stem, self.bn1, self.act1 = model.conv_stem, model.bn1, model.act1
        for i, g in enumerate(groups):
            a, b = g[0], g[1]
            # (a, None): the stage itself; (a, b): nn.Sequential(*blocks[a:b]) -- re-indexed, as the StereoBase / IGEV classes write it;
            # (a, b, "slice"): blocks[a:b] -- nn.Sequential slicing keeps the ORIGINAL indices as keys (lightstereo/backbone.py:47: block3.3.*, block3.4.*)
            blk = model.blocks[a] if b is None else (model.blocks[a:b] if len(g) == 3 else nn.Sequential(*model.blocks[a:b]))
            setattr(self, f"block{i}", blk)
        self._eng = None

    def _trunk(self, x):
        x = self.act1(self.bn1(self.conv_stem(x)))
        outs = []
        for i in range(5):
            x = getattr(self, f"block{i}")(x)
            outs.append(x)
        return outs                                  # x2, x4, x8, x16, x32

    def _trunk_cl(self, img):
        from .. import ops
        stem = cached_pack(self, "_stem", lambda: PackedConv3d(self.conv_stem, self.bn1, ACT_RELU6), mods=(self.conv_stem, self.bn1))
        x = stem(ops.to_cl(img.unsqueeze(2)))          # [N,4,1,H,W], 4th channel zero
        outs = []
        for i in range(5):
            for b in self._leaves(i):
                x = b.forward_cl(x)
            outs.append(x)
        return outs

    def _engine_ok(self, x):
        return on_engine(x) and not self.training and not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))) \
            and x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0 and all(hasattr(b, "forward_cl") for i in range(5) for b in self._leaves(i))

    def _leaves(self, i):
        blk = getattr(self, f"block{i}")
        out = []
        for b in (blk if isinstance(blk, nn.Sequential) else [blk]):
            out += list(b) if isinstance(b, nn.Sequential) else [b]
        return out

    def reset_engine(self):
        for m in self.modules():
            for a in ("_eng", "_stem", "_dec"):
                if a in m.__dict__:
                    m.__dict__[a] = None


class Feature(_Pyramid):
    """models/stereobase/backbone.py:32-73 (`StereoBase.feature`).  output_channels: [48, 64, 192, 160] at 1/4 .. 1/32."""
    output_channels = [48, 64, 192, 160]

    def __init__(self, model=None):
        super().__init__()
        chans = [16, 24, 32, 96, 160]
        self._take_trunk(model if model is not None else create_model(), [(0, 1), (1, 2), (2, 3), (3, 5), (5, 6)])
        from .stereo_models import Conv2xUp
        IN = nn.InstanceNorm2d
        self.deconv32_16 = Conv2xUp(chans[4], chans[3], norm_layer=IN, concat=True)
        self.deconv16_8 = Conv2xUp(chans[3] * 2, chans[2], norm_layer=IN, concat=True)
        self.deconv8_4 = Conv2xUp(chans[2] * 2, chans[1], norm_layer=IN, concat=True)
        self.conv4 = BasicConv2d(chans[1] * 2, chans[1] * 2, norm_layer=IN, act_layer=nn.LeakyReLU, kernel_size=3, stride=1, padding=1)

    def _decoder(self):
        return cached_pack(self, "_dec", lambda: [(_unit(d.conv1), _unit(d.conv2)) for d in (self.deconv32_16, self.deconv16_8, self.deconv8_4)] + [_unit(self.conv4)],
                           mods=(self.deconv32_16, self.deconv16_8, self.deconv8_4, self.conv4))

    def forward_cl(self, img):
        x2, x4, x8, x16, x32 = self._trunk_cl(img)
        d = self._decoder()
        x16 = _up_cat(*d[0], x32, x16, "xr")
        x8 = _up_cat(*d[1], x16, x8, "xr")
        x4 = d[3](_up_cat(*d[2], x8, x4, "xr"))
        return [x4, x8, x16, x32]

    @amp.contract("cast")
    def forward(self, x):
        if self._engine_ok(x):
            return [cl_to_nchw(t, c) for t, c in zip(self.forward_cl(x.float()), self.output_channels)]
        x2, x4, x8, x16, x32 = self._trunk(x)
        x16 = self.deconv32_16(x32, x16)
        x8 = self.deconv16_8(x16, x8)
        x4 = self.conv4(self.deconv8_4(x8, x4))
        return [x4, x8, x16, x32]


class IGEVFeature(Feature):
    """models/igev/extractor.py:320-355 (`IGEVStereo.feature`): the same pyramid with the IGEV unit names (`.conv1.conv` / `.conv1.IN`)."""

    def __init__(self, model=None):
        _Pyramid.__init__(self)
        chans = [16, 24, 32, 96, 160]
        self._take_trunk(model if model is not None else create_model(), [(0, 1), (1, 2), (2, 3), (3, 5), (5, 6)])
        from .stereo_models import Conv2xIGEV, BasicConvIN
        self.deconv32_16 = Conv2xIGEV(chans[4], chans[3], norm="in")
        self.deconv16_8 = Conv2xIGEV(chans[3] * 2, chans[2], norm="in")
        self.deconv8_4 = Conv2xIGEV(chans[2] * 2, chans[1], norm="in")
        self.conv4 = BasicConvIN(chans[1] * 2, chans[1] * 2, kernel_size=3, stride=1, padding=1)


class LightStereoBackbone(_Pyramid):
    """models/lightstereo/backbone.py:29-75 (`LightStereo.backbone`, MobileNetv2 variant).  output_channels [24, 32, 96, 160]."""

    def __init__(self, backbone="MobileNetv2", model=None):
        super().__init__()
        if backbone != "MobileNetv2":
            raise NotImplementedError("LightStereoBackbone: only the MobileNetv2 variant is mirrored (timm is not available offline)")
        channels = [160, 96, 32, 24]
        self._take_trunk(model if model is not None else create_model(), [(0, None), (1, None), (2, None), (3, 5, "slice"), (5, None)])
        from .stereo_models import FPNLayer
        self.fpn_layer4 = FPNLayer(channels[0], channels[1])
        self.fpn_layer3 = FPNLayer(channels[1], channels[2])
        self.fpn_layer2 = FPNLayer(channels[2], channels[3])
        self.out_conv = BasicConv2d(channels[3], channels[3], kernel_size=3, padding=1, padding_mode="replicate", norm_layer=nn.InstanceNorm2d)
        self.output_channels = channels[::-1]

    def _decoder(self):
        return cached_pack(self, "_dec", lambda: [(_unit(f.deconv), _unit(f.conv)) for f in (self.fpn_layer4, self.fpn_layer3, self.fpn_layer2)] + [_unit(self.out_conv)],
                           mods=(self.fpn_layer4, self.fpn_layer3, self.fpn_layer2, self.out_conv))

    def forward_cl(self, img):
        c1, c2, c3, c4, c5 = self._trunk_cl(img)
        d = self._decoder()
        p4 = _up_cat(*d[0], c5, c4, "rx")
        p3 = _up_cat(*d[1], p4, c3, "rx")
        p2 = d[3](_up_cat(*d[2], p3, c2, "rx"))
        return [p2, p3, p4, c5]

    @amp.contract("cast")
    def forward(self, images):
        if self._engine_ok(images):
            return [cl_to_nchw(t, c) for t, c in zip(self.forward_cl(images.float()), self.output_channels)]
        c1, c2, c3, c4, c5 = self._trunk(images)
        p4 = self.fpn_layer4(c5, c4)
        p3 = self.fpn_layer3(p4, c3)
        p2 = self.out_conv(self.fpn_layer2(p3, c2))
        return [p2, p3, p4, c5]
