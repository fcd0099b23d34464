"""MultiBasicEncoder -- the context network in front of the GRU loop of IGEV-Stereo / StereoBase -- on the gfx950 engine.

Mirror of stereo/modeling/models/igev/extractor.py:6-60 (ResidualBlock) and :194-297 (MultiBasicEncoder); the StereoBase copy
(models/stereobase/gru_blocks.py:8-148) is the same network without the `dual_inp` argument.  Same constructor arguments, attribute names
and state_dict keys, so `cnet.*` checkpoint entries load.  Unlike the timm feature pyramids this module is plain PyTorch in the reference
(its file only imports timm), so it is pinned against the reference's own class (tests/golden/context_encoder.npz).

Engine path (eval mode, norm_fn='batch', the configuration every shipped cfg uses): every conv + BatchNorm (+ ReLU) is one fused MFMA
launch on NHWC tensors -- the 7x7 stem at FULL resolution, two full-resolution 64-channel ResidualBlocks, the strided blocks down to
1/16, the five output heads.  A ResidualBlock applies the ReLU to its branch BEFORE the sum, y = relu(x + relu(bn2(conv2(.)))): the
second conv's epilogue does exactly that (OSA_RES_AFTER_ACT).  Other norm functions, dropout and training mode run the reference's torch
composition (with engine convolutions through autograd.engine_convs() when gradients are required)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import amp
from .. import autograd as AG
from .. import ops
from ..engine import cached_pack, foldable_bn, PackedConv3d, ACT_NONE, ACT_RELU
from ..ops import on_engine
from .lightstereo import cl_to_nchw


def _norm(norm_fn, planes, groups=None):
    if norm_fn == "group":
        return nn.GroupNorm(num_groups=groups if groups is not None else planes // 8, num_channels=planes)
    if norm_fn == "batch":
        return nn.BatchNorm2d(planes)
    if norm_fn == "instance":
        return nn.InstanceNorm2d(planes)
    return nn.Sequential()


class ResidualBlock(nn.Module):
    """extractor.py:6-60"""

    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1, self.norm2 = _norm(norm_fn, planes), _norm(norm_fn, planes)
        if stride == 1 and in_planes == planes:
            self.downsample = None
        else:
            self.norm3 = _norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def packs(self):
        bn = foldable_bn          # any _BatchNorm (nn.SyncBatchNorm after convert_sync_batchnorm included); raises on a norm it cannot fold
        return cached_pack(self, "_eng", lambda: (
            PackedConv3d(self.conv1, bn(self.norm1), ACT_RELU), PackedConv3d(self.conv2, bn(self.norm2), ACT_RELU),
            None if self.downsample is None else PackedConv3d(self.downsample[0], bn(self.norm3), ACT_NONE)))

    def forward_cl(self, x):
        c1, c2, ds = self.packs()
        skip = x if ds is None else ds(x)
        return c2(c1(x), residual=skip, res_after_act=True)          # relu(skip + relu(bn2(conv2(relu(bn1(conv1 x))))))

    def forward(self, x):
        y = self.relu(self.norm1(self.conv1(x)))
        y = self.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


class MultiBasicEncoder(nn.Module):
    """extractor.py:194-297 / gru_blocks.py:62-148.  forward(x, dual_inp=False, num_layers=3) -> (outputs04, outputs08, outputs16[, v])."""
    use_engine = True

    def __init__(self, output_dim=None, norm_fn="batch", dropout=0.0, downsample=3):
        super().__init__()
        output_dim = [128] if output_dim is None else output_dim
        self.norm_fn, self.downsample = norm_fn, downsample
        self.norm1 = _norm(norm_fn, 64, groups=8)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=1 + (downsample > 2), padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = 64
        self.layer1 = self._make_layer(64, stride=1)
        self.layer2 = self._make_layer(96, stride=1 + (downsample > 1))
        self.layer3 = self._make_layer(128, stride=1 + (downsample > 0))
        self.layer4 = self._make_layer(128, stride=2)
        self.layer5 = self._make_layer(128, stride=2)
        self.outputs04 = nn.ModuleList([nn.Sequential(ResidualBlock(128, 128, norm_fn, stride=1), nn.Conv2d(128, dim[2], 3, padding=1)) for dim in output_dim])
        self.outputs08 = nn.ModuleList([nn.Sequential(ResidualBlock(128, 128, norm_fn, stride=1), nn.Conv2d(128, dim[1], 3, padding=1)) for dim in output_dim])
        self.outputs16 = nn.ModuleList([nn.Conv2d(128, dim[0], 3, padding=1) for dim in output_dim])
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        for m in self.modules():                                   # extractor.py:262-269
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _make_layer(self, dim, stride=1):
        layers = (ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride), ResidualBlock(dim, dim, self.norm_fn, stride=1))
        self.in_planes = dim
        return nn.Sequential(*layers)

    def reset_engine(self):
        for m in self.modules():
            if hasattr(m, "_eng"):
                m._eng = None
        self._stem = self._heads = None

    # ---- engine path
    def _engine_ok(self, x):
        return self.use_engine and not self.training and self.norm_fn == "batch" and on_engine(x) and \
            not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())))

    def forward_cl(self, x, dual_inp=False, num_layers=3):
        """x: NCHW image batch.  Returns the reference's structure with NHWC engine tensors [B, C, 1, H, W] as leaves."""
        stem = cached_pack(self, "_stem", lambda: PackedConv3d(self.conv1, self.norm1, ACT_RELU), mods=(self.conv1, self.norm1))
        heads = cached_pack(self, "_heads", lambda: ([PackedConv3d(f[1]) for f in self.outputs04], [PackedConv3d(f[1]) for f in self.outputs08],
                                                       [PackedConv3d(f) for f in self.outputs16]),
                            mods=tuple(f[1] for f in self.outputs04) + tuple(f[1] for f in self.outputs08) + tuple(self.outputs16))
        t = stem(ops.to_cl(x.unsqueeze(2)))                          # [N,4,1,H,W] (4th channel zero) -> 64 channels
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                t = blk.forward_cl(t)
        v = t
        if dual_inp:
            t = t[:t.shape[0] // 2]                                  # a batch slice of an NDHWC tensor is an NDHWC tensor
        o04 = [h(f[0].forward_cl(t)) for f, h in zip(self.outputs04, heads[0])]
        if num_layers == 1:
            return (o04, v) if dual_inp else (o04,)
        y = t
        for blk in self.layer4:
            y = blk.forward_cl(y)
        o08 = [h(f[0].forward_cl(y)) for f, h in zip(self.outputs08, heads[1])]
        if num_layers == 2:
            return (o04, o08, v) if dual_inp else (o04, o08)
        z = y
        for blk in self.layer5:
            z = blk.forward_cl(z)
        o16 = [h(z) for h in heads[2]]
        return (o04, o08, o16, v) if dual_inp else (o04, o08, o16)

    def _forward_torch(self, x, dual_inp, num_layers):
        x = self.relu1(self.norm1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        v = x
        if dual_inp:
            x = x[:(x.shape[0] // 2)]
        outputs04 = [f(x) for f in self.outputs04]
        if num_layers == 1:
            return (outputs04, v) if dual_inp else (outputs04,)
        y = self.layer4(x)
        outputs08 = [f(y) for f in self.outputs08]
        if num_layers == 2:
            return (outputs04, outputs08, v) if dual_inp else (outputs04, outputs08)
        z = self.layer5(y)
        outputs16 = [f(z) for f in self.outputs16]
        return (outputs04, outputs08, outputs16, v) if dual_inp else (outputs04, outputs08, outputs16)

    @amp.contract("cast")
    def forward(self, x, dual_inp=False, num_layers=3):
        if self._engine_ok(x):
            def back(t):
                return cl_to_nchw(t, t.shape[1])
            out = self.forward_cl(x, dual_inp, num_layers)
            return tuple([back(t) for t in o] if isinstance(o, list) else back(o) for o in out)
        if on_engine(x) and (self.training or torch.is_grad_enabled()):
            with AG.engine_convs():                                  # training: eligible (stride-1) convolutions forward + backward on the engine
                return self._forward_torch(x, dual_inp, num_layers)
        return self._forward_torch(x, dual_inp, num_layers)
