"""LightStereo 2-D cost aggregation on the gfx950 engine.

Mirror of stereo/modeling/models/lightstereo/aggregation.py:7-134 -- same class names, constructor
arguments and state_dict keys (`conv0.N.pwconv.0.weight`, `att0.conv0_1.weight`, ...) -- with the
forward pass on the engine.  The correlation volume is treated as a [B, D/4, H/4, W/4] feature map
(disparity = channel); every tensor is NHWC between layers:

  * 1x1 expand / project convs (+BN +ReLU6, + the residual of MobileV2Residual) and the two stride-2
    ConvTranspose2d (+BN + redir residual + ReLU): MFMA implicit GEMM (PackedConv3d, D = 1),
  * depthwise 3x3 (stride 1/2) + BN + ReLU6 and the strip convolutions of AttentionModule:
    DepthwiseConv2d (fp32 VALU, HBM/L2 bound); the `attn + attn_0 + attn_1 + attn_2` sum is folded
    into the addend of the second strip conv of each branch (same left-to-right order),
  * `attn * cost` is the raw-gate epilogue of AttentionModule.conv3.

forward() accepts the reference's NCHW tensors and returns `[conv6]` NCHW like the reference;
forward_cl() is the channels-last entry used inside engine chains.  No CPU path.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import amp
from .. import autograd as AG
from ..engine import cached_pack, PackedConv3d, DepthwiseConv2d, ACT_NONE, ACT_RELU, ACT_RELU6
from ..ops import empty_cl, is_cl, on_engine


def nchw_to_cl(x):
    """[B,C,H,W] (any strides) -> logical [B,Cpad4,1,H,W] NDHWC engine tensor (padded channels zero)."""
    B, C, H, W = x.shape
    Cp = (C + 3) // 4 * 4
    out = empty_cl(B, Cp, 1, H, W, x.device)
    if Cp != C:
        out.zero_()
    out[:, :C, 0] = x.float()
    if getattr(x, "_osa_meta", None) is not None:        # same values in another layout: the range block carries over
        out._osa_meta = x._osa_meta
    return out


def cl_to_nchw(x, C=None):
    C = x.shape[1] if C is None else C
    return x[:, :C, 0].contiguous()


F16_CHAIN = os.environ.get("OSA_LS_F16_CHAIN", "1") != "0"     # f16 mode: fp16 tensors between a block's expansion, depthwise and projection layers


class MobileV2Residual(nn.Module):
    """aggregation.py:63-98"""

    def __init__(self, inp, oup, stride, expanse_ratio, dilation=1):
        super().__init__()
        self.stride = stride
        assert stride in [1, 2]
        hidden_dim = int(inp * expanse_ratio)
        self.use_res_connect = self.stride == 1 and inp == oup
        pad = dilation
        self.pwconv = nn.Sequential(nn.Conv2d(inp, hidden_dim, 1, 1, 0, bias=False), nn.BatchNorm2d(hidden_dim), nn.ReLU6(inplace=True))
        self.dwconv = nn.Sequential(nn.Conv2d(hidden_dim, hidden_dim, 3, stride, pad, dilation=dilation, groups=hidden_dim, bias=False),
                                    nn.BatchNorm2d(hidden_dim), nn.ReLU6(inplace=True))
        self.pwliner = nn.Sequential(nn.Conv2d(hidden_dim, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup))
        self._eng = None

    def _pack(self):
        return cached_pack(self, "_eng", lambda: (PackedConv3d(self.pwconv[0], self.pwconv[1], ACT_RELU6),
                                                  DepthwiseConv2d(self.dwconv[0], self.dwconv[1], ACT_RELU6),
                                                  PackedConv3d(self.pwliner[0], self.pwliner[1], ACT_NONE)))

    def forward_cl(self, x):
        pw, dw, pl = self._pack()
        if F16_CHAIN and pw.precision == "f16" and dw.k == (3, 3) and dw.dil == (1, 1) and pw.Co % 8 == 0:
            # f16 mode (r6): the expanded tensor (expanse_ratio x the channels: the bytes that bound these layers) stays fp16 between the three
            # launches, as under the reference's autocast; the block's input / output keep their dtype
            return pl(dw(pw(x, out_split=True), out_f16=True), residual=x if self.use_res_connect else None)
        return pl(dw(pw(x)), residual=x if self.use_res_connect else None)     # x + feat fused in the epilogue

    def forward_train(self, x):
        """aggregation.py:91-98 as a torch composition: the 1x1 convolutions run on the engine (forward, dgrad, wgrad) under
        AG.engine_convs(); depthwise convolutions, BatchNorm (batch statistics) and ReLU6 are torch ops."""
        with AG.engine_convs():
            feat = self.pwliner(self.dwconv(self.pwconv(x)))
        return x + feat if self.use_res_connect else feat

    @amp.contract("res")
    def forward(self, x):
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            return self.forward_train(x)
        return cl_to_nchw(self.forward_cl(nchw_to_cl(x)), self.pwliner[0].out_channels)


class AttentionModule(nn.Module):
    """aggregation.py:101-134"""

    def __init__(self, dim, img_feat_dim):
        super().__init__()
        self.conv0 = nn.Conv2d(img_feat_dim, dim, 1)
        self.conv0_1 = nn.Conv2d(dim, dim, (1, 7), padding=(0, 3), groups=dim)
        self.conv0_2 = nn.Conv2d(dim, dim, (7, 1), padding=(3, 0), groups=dim)
        self.conv1_1 = nn.Conv2d(dim, dim, (1, 11), padding=(0, 5), groups=dim)
        self.conv1_2 = nn.Conv2d(dim, dim, (11, 1), padding=(5, 0), groups=dim)
        self.conv2_1 = nn.Conv2d(dim, dim, (1, 21), padding=(0, 10), groups=dim)
        self.conv2_2 = nn.Conv2d(dim, dim, (21, 1), padding=(10, 0), groups=dim)
        self.conv3 = nn.Conv2d(dim, dim, 1)
        self._eng = None

    def _pack(self):
        return cached_pack(self, "_eng", lambda: dict(
            conv0=PackedConv3d(self.conv0), conv3=PackedConv3d(self.conv3),
            **{n: DepthwiseConv2d(getattr(self, n)) for n in ("conv0_1", "conv0_2", "conv1_1", "conv1_2", "conv2_1", "conv2_2")}))

    def forward_cl(self, cost, x):
        e = self._pack()
        attn = e["conv0"](x)
        s = e["conv0_2"](e["conv0_1"](attn), add=attn)        # attn + attn_0
        s = e["conv1_2"](e["conv1_1"](attn), add=s)           # ... + attn_1
        s = e["conv2_2"](e["conv2_1"](attn), add=s)           # ... + attn_2
        B, C, _, H, W = cost.shape
        gate = cost.permute(0, 2, 3, 4, 1).reshape(B, H, W, C)  # NHWC view of the same memory
        return e["conv3"](s, gate=gate, gate_raw=True)         # conv3(attn) * cost

    def forward_train(self, cost, x):
        """aggregation.py:120-134 (1x1 convs on the engine, strip depthwise convs in torch)"""
        with AG.engine_convs():
            attn = self.conv0(x)
            attn = attn + self.conv0_2(self.conv0_1(attn)) + self.conv1_2(self.conv1_1(attn)) + self.conv2_2(self.conv2_1(attn))
            return self.conv3(attn) * cost

    @amp.contract("gru")                 # conv3(attn) * cost: promoted against the cost's dtype (aggregation.py:45)
    def forward(self, cost, x):
        if self.training or (torch.is_grad_enabled() and (cost.requires_grad or x.requires_grad)):
            return self.forward_train(cost, x)
        return cl_to_nchw(self.forward_cl(nchw_to_cl(cost), nchw_to_cl(x)), self.conv3.out_channels)


class Aggregation(nn.Module):
    """aggregation.py:7-60"""

    def __init__(self, in_channels, left_att, blocks, expanse_ratio, backbone_channels):
        super().__init__()
        self.left_att = left_att
        self.expanse_ratio = expanse_ratio
        c = in_channels
        self.conv0 = nn.Sequential(*[MobileV2Residual(c, c, stride=1, expanse_ratio=expanse_ratio) for _ in range(blocks[0])])
        self.conv1 = MobileV2Residual(c, c * 2, stride=2, expanse_ratio=expanse_ratio)
        self.conv2 = nn.Sequential(*[MobileV2Residual(c * 2, c * 2, stride=1, expanse_ratio=expanse_ratio) for _ in range(blocks[1] - 1)])
        self.conv3 = MobileV2Residual(c * 2, c * 4, stride=2, expanse_ratio=expanse_ratio)
        self.conv4 = nn.Sequential(*[MobileV2Residual(c * 4, c * 4, stride=1, expanse_ratio=expanse_ratio) for _ in range(blocks[2] - 1)])
        self.conv5 = nn.Sequential(nn.ConvTranspose2d(c * 4, c * 2, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm2d(c * 2))
        self.conv6 = nn.Sequential(nn.ConvTranspose2d(c * 2, c, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm2d(c))
        self.redir1 = MobileV2Residual(c, c, stride=1, expanse_ratio=expanse_ratio)
        self.redir2 = MobileV2Residual(c * 2, c * 2, stride=1, expanse_ratio=expanse_ratio)
        if self.left_att:
            self.att0 = AttentionModule(c, backbone_channels[0])
            self.att2 = AttentionModule(c * 2, backbone_channels[1])
            self.att4 = AttentionModule(c * 4, backbone_channels[2])
        self._eng = None

    def reset_engine(self):
        """Drop packed weights (call after loading a checkpoint)."""
        self._eng = None
        for m in self.modules():
            if m is not self and hasattr(m, "_eng"):
                m._eng = None

    def _pack(self):
        return cached_pack(self, "_eng", lambda: (PackedConv3d(self.conv5[0], self.conv5[1], ACT_RELU),
                                                  PackedConv3d(self.conv6[0], self.conv6[1], ACT_RELU)),
                           mods=(self.conv5, self.conv6))

    def forward_cl(self, x, features_left):
        """x: NHWC volume (logical [B,D4,1,H4,W4]); features_left: NHWC maps at 1/4, 1/8, 1/16."""
        assert is_cl(x)
        d5, d6 = self._pack()
        for blk in self.conv0:
            x = blk.forward_cl(x)
        if self.left_att:
            x = self.att0.forward_cl(x, features_left[0])
        conv2 = self.conv1.forward_cl(x)
        for blk in self.conv2:
            conv2 = blk.forward_cl(conv2)
        if self.left_att:
            conv2 = self.att2.forward_cl(conv2, features_left[1])
        conv4 = self.conv3.forward_cl(conv2)
        for blk in self.conv4:
            conv4 = blk.forward_cl(conv4)
        if self.left_att:
            conv4 = self.att4.forward_cl(conv4, features_left[2])
        conv5 = d5(conv4, residual=self.redir2.forward_cl(conv2))     # relu(conv5(conv4) + redir2(conv2))
        conv6 = d6(conv5, residual=self.redir1.forward_cl(x))         # relu(conv6(conv5) + redir1(x))
        return conv6

    def forward_train(self, x, features_left):
        """aggregation.py:44-60; sub-modules take their own training paths (engine 1x1 convs + torch depthwise / BN / ReLU6);
        the two ConvTranspose2d are torch (MIOpen) ops in training."""
        x = self.conv0(x)
        if self.left_att:
            x = self.att0(x, features_left[0])
        conv2 = self.conv2(self.conv1(x))
        if self.left_att:
            conv2 = self.att2(conv2, features_left[1])
        conv4 = self.conv4(self.conv3(conv2))
        if self.left_att:
            conv4 = self.att4(conv4, features_left[2])
        conv5 = F.relu(self.conv5(conv4) + self.redir2(conv2))
        return [F.relu(self.conv6(conv5) + self.redir1(x))]

    @amp.contract("gru")                 # relu(conv6(conv5) + redir1(x)): redir1 is a skip block -> promoted against x's dtype (aggregation.py:58-60)
    def forward(self, x, features_left):
        if not on_engine(x):
            raise RuntimeError("openstereo_amd Aggregation runs on the GPU engine only (no CPU path)")
        if self.training or (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self.forward_train(x, features_left)
        out = self.forward_cl(nchw_to_cl(x), [nchw_to_cl(f) for f in features_left[:3]])
        return [cl_to_nchw(out, self.conv6[0].out_channels)]


class LightStereoCostStage(nn.Module):
    """The volume -> aggregation -> initial-disparity slice of lightstereo.py:51-56 with the reference's
    attribute name (`cost_agg`), so those checkpoint keys load:
        correlation_volume(features_left[0], features_right[0], max_disp // 4)        (a4)
        -> Aggregation(in_channels = max_disp // 4 = 48, ...)                         (a9)
        -> softmax over the disparity channels + disparity_regression                 (a12)
    All three stages on the engine; the quarter-resolution disparity is what `context_upsample` (a13) consumes."""

    def __init__(self, max_disp=192, left_att=True, blocks=(1, 2, 4), expanse_ratio=4, backbone_channels=(24, 32, 96, 160)):
        super().__init__()
        self.max_disp = max_disp
        self.cost_agg = Aggregation(in_channels=max_disp // 4, left_att=left_att, blocks=list(blocks),
                                    expanse_ratio=expanse_ratio, backbone_channels=list(backbone_channels))

    def forward(self, features_left, feature_right):
        from .. import ops
        if not on_engine(features_left[0]):
            raise RuntimeError("openstereo_amd LightStereoCostStage runs on the GPU engine only (no CPU path)")
        D4 = self.max_disp // 4
        vol = ops.correlation_volume(features_left[0], feature_right, D4)              # [B, D/4, H/4, W/4]
        enc = self.cost_agg.forward_cl(nchw_to_cl(vol), [nchw_to_cl(f) for f in features_left[:3]])
        cost = cl_to_nchw(enc, D4)                                                     # squeezed_encoding
        init_disp, prob = ops.softmax_disparity_regression(cost, D4, keepdim=True, return_prob=True)
        return {"init_disp": init_disp, "prob": prob, "encoding_volume": cost}
