"""StereoBase / IGEV cost aggregation on the gfx950 engine.

Both models use the same 3-level 3-D hourglass (stereo/modeling/models/stereobase/hourglass.py:7-104,
models/igev/igev_stereo.py:7-76): Conv3d+BN+LeakyReLU(0.01) pairs with stride 2, k4/s2/p1 transposed
convs, channel concats followed by 1x1x1 convs, and FeatureAtt channel gating
(cv = sigmoid(Conv2d(feat))[:, :, None] * cv) at five points.  They differ only in parameter naming
(`.block.0/.block.1` vs `.conv/.bn`), so two thin module trees share one engine forward:

  * every Conv3d/ConvTranspose3d (+BN +LeakyReLU) is a PackedConv3d (fp32 MFMA implicit GEMM),
  * the gate is fused into the producing conv's epilogue (logits in, sigmoid in-kernel),
  * torch.cat never happens: producers write channel slices of one NDHWC buffer.
The 1x1 2-D convs that produce the gate logits run as ordinary PyTorch-ROCm modules (they belong to
the 2-D feature side).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import amp, ops
from ..engine import cached_pack, foldable_bn, PackedConv3d, SmallCoConv3d, ACT_NONE, ACT_LEAKY, ACT_RELU
from ..ops import empty_cl


# ----------------------------------------------------------------------------- StereoBase-style blocks (.block.N names)
class BasicConv2d(nn.Module):
    """common/basic_block_2d.py:6-21"""

    def __init__(self, cin, cout, kernel_size=3, stride=1, padding=0, bias=False, norm_layer=None, act_layer=None, **kw):
        super().__init__()
        layers = [nn.Conv2d(cin, cout, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias, **kw)]
        if norm_layer is not None:
            layers.append(norm_layer(cout))
        if act_layer is not None:
            layers.append(act_layer())
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        return self.block(x)


class BasicConv3d(nn.Module):
    """common/basic_block_3d.py:5-20 (module tree only; compute goes through PackedConv3d)."""

    def __init__(self, cin, cout, kernel_size=3, stride=1, padding=0, bias=False, norm_layer=None, act_layer=None, **kw):
        super().__init__()
        layers = [nn.Conv3d(cin, cout, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias, **kw)]
        if norm_layer is not None:
            layers.append(norm_layer(cout))
        if act_layer is not None:
            layers.append(act_layer())
        self.block = nn.Sequential(*layers)


class BasicDeconv3d(nn.Module):
    """common/basic_block_3d.py:23-38"""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias=False, norm_layer=None, act_layer=None, **kw):
        super().__init__()
        layers = [nn.ConvTranspose3d(cin, cout, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias, **kw)]
        if norm_layer is not None:
            layers.append(norm_layer(cout))
        if act_layer is not None:
            layers.append(act_layer())
        self.block = nn.Sequential(*layers)


class FeatureAtt(nn.Module):
    """stereobase/igev_blocks.py:35-48.  logits(feat) -> NHWC gate logits for the conv epilogue."""

    def __init__(self, cv_chan, feat_chan):
        super().__init__()
        self.feat_att = nn.Sequential(
            BasicConv2d(feat_chan, feat_chan // 2, norm_layer=nn.BatchNorm2d, act_layer=nn.LeakyReLU,
                        kernel_size=1, stride=1, padding=0),
            nn.Conv2d(feat_chan // 2, cv_chan, 1))

    def logits(self, feat):
        return self.feat_att(feat).permute(0, 2, 3, 1).contiguous().float()


def _pack_sb(m):
    """BasicConv3d/BasicDeconv3d -> PackedConv3d (BN and LeakyReLU(0.01) fused when present)."""
    conv = m.block[0]
    rest = list(m.block)[1:]
    norms = [l for l in rest if not isinstance(l, (nn.LeakyReLU, nn.ReLU))]
    acts = [l for l in rest if isinstance(l, (nn.LeakyReLU, nn.ReLU))]
    assert len(norms) <= 1 and len(acts) <= 1, "BasicConv3d / BasicDeconv3d block = [conv, norm?, act?]"
    bn = foldable_bn(norms[0]) if norms else None        # nn.SyncBatchNorm is a _BatchNorm but not a BatchNorm3d; unknown norms raise
    if acts and isinstance(acts[0], nn.LeakyReLU):
        return PackedConv3d(conv, bn, ACT_LEAKY, acts[0].negative_slope)
    return PackedConv3d(conv, bn, ACT_RELU if acts else ACT_NONE)


# ----------------------------------------------------------------------------- shared engine forward
def hourglass_forward_cl(L, x, gates, return_multi=False):
    """L: dict of PackedConv3d (c1a c1b c2a c2b c3a c3b c3up c2up c1up a0a a0b a0c a1a a1b a1c);
    gates: NHWC logits (g8, g16, g32, g16u, g8u); x: NDHWC [B,c,D,H,W]."""
    B, _, D, H, W = x.shape
    c2ch, c4ch = L["c1b"].Co, L["c2b"].Co
    dev = x.device
    d1 = L["c1a"].out_shape(D, H, W)
    d2 = L["c2a"].out_shape(*d1)
    cat1 = empty_cl(B, 2 * c2ch, *d1, dev)      # [conv2_up | conv1]
    cat2 = empty_cl(B, 2 * c4ch, *d2, dev)      # [conv3_up | conv2]
    L["c1b"](L["c1a"](x), gate=gates["g8"], out=cat1, out_off=c2ch)                          # conv1 + att_8
    L["c2b"](L["c2a"](cat1, x_off=c2ch), gate=gates["g16"], out=cat2, out_off=c4ch)          # conv2 + att_16
    c3 = L["c3b"](L["c3a"](cat2, x_off=c4ch), gate=gates["g32"])                             # conv3 + att_32
    L["c3up"](c3, out=cat2, out_off=0)
    conv2 = L["a0c"](L["a0b"](L["a0a"](cat2)), gate=gates["g16u"])                           # agg_0 + att_up_16
    L["c2up"](conv2, out=cat1, out_off=0)
    conv1 = L["a1c"](L["a1b"](L["a1a"](cat1)), gate=gates["g8u"])                            # agg_1 + att_up_8
    conv = L["c1up"](conv1)
    return [conv, conv1, conv2] if return_multi else conv


class Hourglass(nn.Module):
    """models/stereobase/hourglass.py:7-104 (same parameter names)."""

    def __init__(self, in_channels, backbone_channels=None):
        super().__init__()
        if backbone_channels is None:
            backbone_channels = [48, 64, 192, 120]
        c = in_channels
        cb = lambda i, o, k, p, s: BasicConv3d(i, o, norm_layer=nn.BatchNorm3d, act_layer=nn.LeakyReLU,
                                               kernel_size=k, padding=p, stride=s)
        up = lambda i, o, norm, act: BasicDeconv3d(i, o, norm_layer=norm, act_layer=act, kernel_size=(4, 4, 4),
                                                   padding=(1, 1, 1), stride=(2, 2, 2))
        self.conv1 = nn.Sequential(cb(c, 2 * c, 3, 1, 2), cb(2 * c, 2 * c, 3, 1, 1))
        self.conv2 = nn.Sequential(cb(2 * c, 4 * c, 3, 1, 2), cb(4 * c, 4 * c, 3, 1, 1))
        self.conv3 = nn.Sequential(cb(4 * c, 6 * c, 3, 1, 2), cb(6 * c, 6 * c, 3, 1, 1))
        self.conv3_up = up(6 * c, 4 * c, nn.BatchNorm3d, nn.LeakyReLU)
        self.conv2_up = up(4 * c, 2 * c, nn.BatchNorm3d, nn.LeakyReLU)
        self.conv1_up = up(2 * c, c, None, None)
        self.agg_0 = nn.Sequential(cb(8 * c, 4 * c, 1, 0, 1), cb(4 * c, 4 * c, 3, 1, 1), cb(4 * c, 4 * c, 3, 1, 1))
        self.agg_1 = nn.Sequential(cb(4 * c, 2 * c, 1, 0, 1), cb(2 * c, 2 * c, 3, 1, 1), cb(2 * c, 2 * c, 3, 1, 1))
        self.feature_att_8 = FeatureAtt(2 * c, backbone_channels[1])
        self.feature_att_16 = FeatureAtt(4 * c, backbone_channels[2])
        self.feature_att_32 = FeatureAtt(6 * c, backbone_channels[3])
        self.feature_att_up_16 = FeatureAtt(4 * c, backbone_channels[2])
        self.feature_att_up_8 = FeatureAtt(2 * c, backbone_channels[1])
        self._packed = None

    def reset_engine(self):
        self._packed = None

    def _pack(self):
        P = _pack_sb
        return cached_pack(self, "_packed", lambda: dict(
            c1a=P(self.conv1[0]), c1b=P(self.conv1[1]), c2a=P(self.conv2[0]), c2b=P(self.conv2[1]),
            c3a=P(self.conv3[0]), c3b=P(self.conv3[1]), c3up=P(self.conv3_up), c2up=P(self.conv2_up),
            c1up=P(self.conv1_up), a0a=P(self.agg_0[0]), a0b=P(self.agg_0[1]), a0c=P(self.agg_0[2]),
            a1a=P(self.agg_1[0]), a1b=P(self.agg_1[1]), a1c=P(self.agg_1[2])),
            mods=(self.conv1, self.conv2, self.conv3, self.conv3_up, self.conv2_up, self.conv1_up, self.agg_0, self.agg_1))

    def gate_logits(self, features):
        return dict(g8=self.feature_att_8.logits(features[1]), g16=self.feature_att_16.logits(features[2]),
                    g32=self.feature_att_32.logits(features[3]), g16u=self.feature_att_up_16.logits(features[2]),
                    g8u=self.feature_att_up_8.logits(features[1]))

    def forward_cl(self, x, features, return_multi=False):
        return hourglass_forward_cl(self._pack(), x, self.gate_logits(features), return_multi)

    @staticmethod
    def _unit_train(m, x):
        """BasicConv3d / BasicDeconv3d in training mode: the convolution (forward, dgrad, wgrad) on the
        engine through autograd, BatchNorm3d / LeakyReLU as the reference's own torch modules."""
        from .. import autograd as A
        x = A.conv_module(m.block[0], x)
        for layer in list(m.block)[1:]:
            x = layer(x)
        return x

    def forward_train(self, x, features, return_multi=False):
        """hourglass.py:79-104 with differentiable engine convolutions (BASELINE configs[2]: StereoBase
        training).  FeatureAtt gates, concatenations, BatchNorm and activations are torch ops so that
        batch statistics, SyncBN and DDP behave exactly like the reference."""
        seq = lambda mods, t: [t := self._unit_train(m, t) for m in mods][-1]
        att = lambda fa, cv, feat: torch.sigmoid(fa.feat_att(feat).unsqueeze(2)) * cv
        conv1 = att(self.feature_att_8, seq(self.conv1, x), features[1])
        conv2 = att(self.feature_att_16, seq(self.conv2, conv1), features[2])
        conv3 = att(self.feature_att_32, seq(self.conv3, conv2), features[3])
        conv2 = torch.cat((self._unit_train(self.conv3_up, conv3), conv2), dim=1)
        conv2 = att(self.feature_att_up_16, seq(self.agg_0, conv2), features[2])
        conv1 = torch.cat((self._unit_train(self.conv2_up, conv2), conv1), dim=1)
        conv1 = att(self.feature_att_up_8, seq(self.agg_1, conv1), features[1])
        conv = self._unit_train(self.conv1_up, conv1)
        return [conv, conv1, conv2] if return_multi else conv

    @amp.contract("cast")
    def forward(self, x, features, return_multi=False):
        """Drop-in: NCDHW in -> NCDHW out.  Training mode (or grad-requiring inputs) takes the autograd path."""
        if self.training or (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self.forward_train(x, features, return_multi)
        out = self.forward_cl(ops.to_cl(x), features, return_multi)
        if return_multi:
            return [ops.to_ncdhw(t) for t in out]
        return ops.to_ncdhw(out, channels=x.shape[1])


class StereoBaseCostStage(nn.Module):
    """The volume -> aggregation -> initial-disparity slice of stereobase_gru.py:139-164 with the
    reference's attribute names (`cost_agg`, `classifier`), so those checkpoint keys load:
      [gwc volume (num_groups)] [+ concat volume] [+ extra volumes] -> Hourglass -> Conv3d(c,1,3) -> softmax -> regression.
    `extra_channels` / `extras`: the dormant variants of stereobase_gru.py:152-159 (`build_sub_volume`: 1 channel, `InterlacedVolume`:
    INTERLACED_CHANNELS) arrive as NCDHW tensors and are appended behind the fused gwc + concat channels, in the reference's order."""

    def __init__(self, max_disp=192, num_groups=8, concat_channels=8, backbone_channels=None, extra_channels=0):
        super().__init__()
        self.max_disp, self.num_groups, self.concat_channels, self.extra_channels = max_disp, num_groups, concat_channels, extra_channels
        volume_channel = num_groups + 2 * concat_channels + extra_channels
        self.cost_agg = Hourglass(volume_channel, backbone_channels)
        self.classifier = nn.Conv3d(volume_channel, 1, 3, 1, 1, bias=False)
        self._cls = None

    def reset_engine(self):
        self._cls = None
        self.cost_agg.reset_engine()

    def forward_train(self, match_left, match_right, concat_left, concat_right, features_left, extras=None):
        """Training path (BASELINE configs[2]): differentiable engine ops end to end -- volumes, hourglass
        convolutions, classifier, fused softmax + regression -- BatchNorm / activations as torch modules."""
        from .. import autograd as A
        D4 = self.max_disp // 4
        parts = []
        if self.num_groups:
            parts.append(A.build_gwc_volume(match_left, match_right, D4, self.num_groups))
        if self.concat_channels:
            parts.append(A.build_concat_volume(concat_left, concat_right, D4))
        parts += [e.float() for e in (extras or ())]
        vol = torch.cat(parts, 1)
        geo = self.cost_agg.forward_train(vol, features_left)
        cost = A.conv_module(self.classifier, geo).squeeze(1)
        init_disp = A.softmax_disparity_regression(cost, keepdim=True)
        return {"init_disp": init_disp, "prob": torch.softmax(cost, dim=1), "geo_encoding_volume": geo}

    def forward(self, match_left, match_right, concat_left, concat_right, features_left, extras=None):
        if self.training or (torch.is_grad_enabled() and (match_left.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self.forward_train(match_left, match_right, concat_left, concat_right, features_left, extras)
        assert sum(e.shape[1] for e in (extras or ())) == self.extra_channels, "extras do not match extra_channels"
        D4 = self.max_disp // 4
        if (self.num_groups + 2 * self.concat_channels + self.extra_channels) % 4:
            # the fused NDHWC chain (concatenations at channel offsets c, 2c, 4c) needs channel counts in multiples of 4; the dormant
            # configurations that break this (a 1-channel sub volume: 33 channels) take the composition of the training path instead --
            # the same engine convolutions through their autograd Functions, BatchNorm / activations as torch modules
            with torch.no_grad():
                return self.forward_train(match_left, match_right, concat_left, concat_right, features_left, extras)
        vol = None
        if self.num_groups or self.concat_channels:
            vol = ops.build_cost_volume_cl(match_left if self.num_groups else None, match_right if self.num_groups else None, self.num_groups,
                                           concat_left if self.concat_channels else None, concat_right if self.concat_channels else None, maxdisp=D4)
        if self.extra_channels:
            # dormant variants: NCDHW pieces copied behind the fused channels of one NDHWC buffer (strided torch copies: not a tuned path)
            n0 = self.num_groups + 2 * self.concat_channels
            B, _, H, W = match_left.shape
            full = ops.empty_cl(B, (n0 + self.extra_channels + 3) // 4 * 4, D4, H, W, match_left.device)
            full.zero_()
            if vol is not None:
                full[:, :n0] = vol[:, :n0]
            c = n0
            for e in extras:
                full[:, c:c + e.shape[1]] = e.float()
                c += e.shape[1]
            vol = full
        geo = self.cost_agg.forward_cl(vol, features_left)
        cost = cached_pack(self, "_cls", lambda: SmallCoConv3d(self.classifier), mods=(self.classifier,))(geo)   # [B,1,D/4,H/4,W/4]
        init_disp, prob = ops.softmax_disparity_regression(cost[:, 0], self.max_disp // 4, keepdim=True, return_prob=True)
        return {"init_disp": init_disp, "prob": prob, "geo_encoding_volume": geo}


# ----------------------------------------------------------------------------- IGEV naming (.conv/.bn)
class BasicConv(nn.Module):
    """models/igev/submodule.py:6-32 (LeakyReLU(0.01) when relu=True)."""

    def __init__(self, cin, cout, deconv=False, is_3d=False, bn=True, relu=True, **kw):
        super().__init__()
        self.relu, self.use_bn = relu, bn
        if is_3d:
            self.conv = (nn.ConvTranspose3d if deconv else nn.Conv3d)(cin, cout, bias=False, **kw)
            self.bn = nn.BatchNorm3d(cout)
        else:
            self.conv = (nn.ConvTranspose2d if deconv else nn.Conv2d)(cin, cout, bias=False, **kw)
            self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):            # only used for the 2-D gate branch
        x = self.conv(x)
        if self.use_bn:
            x = self.bn(x)
        return nn.functional.leaky_relu(x, 0.01) if self.relu else x


class IGEVFeatureAtt(nn.Module):
    """models/igev/submodule.py:237-250"""

    def __init__(self, cv_chan, feat_chan):
        super().__init__()
        self.feat_att = nn.Sequential(BasicConv(feat_chan, feat_chan // 2, kernel_size=1, stride=1, padding=0),
                                      nn.Conv2d(feat_chan // 2, cv_chan, 1))

    def logits(self, feat):
        return self.feat_att(feat).permute(0, 2, 3, 1).contiguous().float()


def _pack_igev(m):
    return PackedConv3d(m.conv, m.bn if m.use_bn else None, ACT_LEAKY if m.relu else ACT_NONE, 0.01)


class hourglass(nn.Module):
    """models/igev/igev_stereo.py:7-76 (same parameter names; conv1_up always emits 8 channels)."""

    def __init__(self, in_channels):
        super().__init__()
        c = in_channels
        cb = lambda i, o, k, p, s: BasicConv(i, o, is_3d=True, bn=True, relu=True, kernel_size=k, padding=p, stride=s, dilation=1)
        up = lambda i, o, bn, relu: BasicConv(i, o, deconv=True, is_3d=True, bn=bn, relu=relu, kernel_size=(4, 4, 4),
                                              padding=(1, 1, 1), stride=(2, 2, 2))
        ag = lambda i, o, k, p: BasicConv(i, o, is_3d=True, kernel_size=k, padding=p, stride=1)
        self.conv1 = nn.Sequential(cb(c, 2 * c, 3, 1, 2), cb(2 * c, 2 * c, 3, 1, 1))
        self.conv2 = nn.Sequential(cb(2 * c, 4 * c, 3, 1, 2), cb(4 * c, 4 * c, 3, 1, 1))
        self.conv3 = nn.Sequential(cb(4 * c, 6 * c, 3, 1, 2), cb(6 * c, 6 * c, 3, 1, 1))
        self.conv3_up = up(6 * c, 4 * c, True, True)
        self.conv2_up = up(4 * c, 2 * c, True, True)
        self.conv1_up = up(2 * c, 8, False, False)
        self.agg_0 = nn.Sequential(ag(8 * c, 4 * c, 1, 0), ag(4 * c, 4 * c, 3, 1), ag(4 * c, 4 * c, 3, 1))
        self.agg_1 = nn.Sequential(ag(4 * c, 2 * c, 1, 0), ag(2 * c, 2 * c, 3, 1), ag(2 * c, 2 * c, 3, 1))
        self.feature_att_8 = IGEVFeatureAtt(2 * c, 64)
        self.feature_att_16 = IGEVFeatureAtt(4 * c, 192)
        self.feature_att_32 = IGEVFeatureAtt(6 * c, 160)
        self.feature_att_up_16 = IGEVFeatureAtt(4 * c, 192)
        self.feature_att_up_8 = IGEVFeatureAtt(2 * c, 64)
        self._packed = None

    def reset_engine(self):
        self._packed = None

    def _packed_layers(self):
        P = _pack_igev
        return cached_pack(self, "_packed", lambda: dict(
            c1a=P(self.conv1[0]), c1b=P(self.conv1[1]), c2a=P(self.conv2[0]), c2b=P(self.conv2[1]),
            c3a=P(self.conv3[0]), c3b=P(self.conv3[1]), c3up=P(self.conv3_up), c2up=P(self.conv2_up),
            c1up=P(self.conv1_up), a0a=P(self.agg_0[0]), a0b=P(self.agg_0[1]), a0c=P(self.agg_0[2]),
            a1a=P(self.agg_1[0]), a1b=P(self.agg_1[1]), a1c=P(self.agg_1[2])),
            mods=(self.conv1, self.conv2, self.conv3, self.conv3_up, self.conv2_up, self.conv1_up, self.agg_0, self.agg_1))

    def forward_cl(self, x, features):
        gates = dict(g8=self.feature_att_8.logits(features[1]), g16=self.feature_att_16.logits(features[2]),
                     g32=self.feature_att_32.logits(features[3]), g16u=self.feature_att_up_16.logits(features[2]),
                     g8u=self.feature_att_up_8.logits(features[1]))
        return hourglass_forward_cl(self._packed_layers(), x, gates)

    @staticmethod
    def _unit_train(m, x):
        """BasicConv (submodule.py:6-32) in training mode: the convolution on the engine through autograd, BatchNorm3d / LeakyReLU as torch ops."""
        from .. import autograd as A
        x = A.conv_module(m.conv, x)
        if m.use_bn:
            x = m.bn(x)
        return nn.functional.leaky_relu(x, 0.01) if m.relu else x

    def forward_train(self, x, features):
        """igev_stereo.py:51-76 with differentiable engine convolutions; FeatureAtt gates, concatenations, BatchNorm and activations are
        torch ops (batch statistics, SyncBN and DDP behave like the reference)."""
        u = self._unit_train
        seq = lambda mods, t: [t := u(m, t) for m in mods][-1]
        att = lambda fa, cv, feat: torch.sigmoid(fa.feat_att(feat).unsqueeze(2)) * cv
        conv1 = att(self.feature_att_8, seq(self.conv1, x), features[1])
        conv2 = att(self.feature_att_16, seq(self.conv2, conv1), features[2])
        conv3 = att(self.feature_att_32, seq(self.conv3, conv2), features[3])
        conv2 = torch.cat((u(self.conv3_up, conv3), conv2), dim=1)
        conv2 = att(self.feature_att_up_16, seq(self.agg_0, conv2), features[2])
        conv1 = torch.cat((u(self.conv2_up, conv2), conv1), dim=1)
        conv1 = att(self.feature_att_up_8, seq(self.agg_1, conv1), features[1])
        return u(self.conv1_up, conv1)

    @amp.contract("cast")
    def forward(self, x, features):
        if self.training or (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self.forward_train(x, features)
        return ops.to_ncdhw(self.forward_cl(ops.to_cl(x), features), channels=8)
