"""InterlacedVolume (cost_volume.py:120-169) -- a dormant volume variant of StereoBase (`USE_INTERLACED_VOLUME`, no shipped config turns it
on), built so that the option is honoured instead of refused.

What the reference computes, per disparity i: crop the left features to x >= i and the right features to x < W - i, interweave their
channels (left at even, right at odd depth positions: 2 x 96 = 192), view the result as a ONE-channel volume [B,1,192,H,W-i] and run
  Conv3d(1, 16, (8,3,3), stride (8,1,1)) -> Conv3d(16, 32, (8,3,3), stride (8,1,1)) -> Conv3d(32, 16, (3,3,3), stride (3,1,1))
(each + BatchNorm3d + ReLU; depth 192 -> 24 -> 3 -> 1), then a 1x1 Conv2d(16, num_features) + BatchNorm2d + ReLU; the result is plane i of
the volume for x >= i (zero elsewhere).

The depth stride of every Conv3d equals its depth kernel, so none of them is a 3-D convolution: it is a 2-D 3x3 convolution over
(kd x Cin) channels applied to each depth group independently with SHARED weights -- groups are batch items.  With the depth taps
outermost in the channel index (kd * Cin + c) the regrouping between stages is a pure reshape:
    [B,192,H,W'] = [B*24, 8, H, W']      --conv2d   8 -> 16-->  [B*24, 16, H, W'] = [B*3, 8*16, H, W']
                                          --conv2d 128 -> 32-->  [B*3, 32, H, W']  = [B, 3*32, H, W']
                                          --conv2d  96 -> 16-->  [B, 16, H, W']  --1x1 16 -> F-->  [B, F, H, W']
In eval mode on the GPU these four layers are fused conv + BN + ReLU launches of the engine's 2-D MFMA kernels, per disparity on the
cropped width (the crop's zero padding is part of the semantics: column x = i sees zeros at x = i - 1, not the neighbouring pixel).
In training mode the module is the reference's torch composition."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..engine import PackedConv3d, cached_pack, ACT_RELU
from ..ops import to_cl, to_ncdhw, on_engine
from .igev_style import BasicConv2d, BasicConv3d


class InterlacedVolume(nn.Module):
    def __init__(self, num_features=8):
        super().__init__()
        self.num_features = num_features
        c3 = lambda i, o, k, s: BasicConv3d(i, o, norm_layer=nn.BatchNorm3d, act_layer=nn.ReLU, kernel_size=k, stride=s, padding=(0, 1, 1))
        self.conv3d = nn.Sequential(c3(1, 16, (8, 3, 3), (8, 1, 1)), c3(16, 32, (8, 3, 3), (8, 1, 1)), c3(32, 16, (3, 3, 3), (3, 1, 1)))
        self.volume11 = BasicConv2d(16, num_features, norm_layer=nn.BatchNorm2d, act_layer=nn.ReLU, kernel_size=1, stride=1)
        self._eng = None

    def reset_engine(self):
        self._eng = None

    @staticmethod
    def interweave_tensors(refimg_fea, targetimg_fea):
        B, C, H, W = refimg_fea.shape
        return torch.stack((refimg_fea, targetimg_fea), 2).reshape(B, 2 * C, H, W)       # left at even, right at odd channels

    def _packs(self):
        def flat(block):
            """Conv3d whose depth stride equals its depth kernel -> the 2-D layer over kd * Cin channels (depth tap outermost)."""
            conv, bn = block.block[0], block.block[1]
            Co, Ci, kd = conv.weight.shape[:3]
            c2 = nn.Conv2d(Ci * kd, Co, 3, padding=1, bias=False).to(conv.weight.device)
            c2.weight.data = conv.weight.detach().permute(0, 2, 1, 3, 4).reshape(Co, kd * Ci, 3, 3).contiguous()
            return PackedConv3d(c2, bn, ACT_RELU)
        return [flat(b) for b in self.conv3d] + [PackedConv3d(self.volume11.block[0], self.volume11.block[1], ACT_RELU)]

    def _forward_torch(self, feat_l, feat_r, maxdisp):
        B, C, H, W = feat_l.shape
        volume = feat_l.new_zeros([B, self.num_features, maxdisp, H, W])
        for i in range(maxdisp):
            x = self.interweave_tensors(feat_l[:, :, :, i:], feat_r[:, :, :, :W - i])
            x = x.unsqueeze(1)
            for blk in self.conv3d:                  # BasicConv3d mirrors are module trees (their compute normally goes through PackedConv3d)
                x = blk.block(x)
            x = self.volume11(x.squeeze(2))
            volume[:, :, i, :, i:] = x
        return volume.contiguous()

    def forward(self, feat_l, feat_r, maxdisp):
        B, C, H, W = feat_l.shape
        if 2 * C != 192:
            raise ValueError(f"InterlacedVolume: the (8,8,3) depth kernels reduce exactly 2 x 96 interwoven channels to one plane; got C = {C}")
        if self.training or (torch.is_grad_enabled() and (feat_l.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self._forward_torch(feat_l, feat_r, maxdisp)
        if not on_engine(feat_l):
            raise RuntimeError("openstereo_amd InterlacedVolume runs on the GPU engine only (no CPU path)")
        e = cached_pack(self, "_eng", self._packs)
        fl, fr = feat_l.float(), feat_r.float()
        volume = fl.new_zeros([B, self.num_features, maxdisp, H, W])
        for i in range(min(maxdisp, W)):
            Wc = W - i
            x = self.interweave_tensors(fl[:, :, :, i:], fr[:, :, :, :Wc])                     # [B,192,H,Wc]
            x = to_cl(x.reshape(B * 24, 8, 1, H, Wc))
            x = e[0](x)                                                                          # [B*24,16,1,H,Wc] NHWC
            x = to_cl(to_ncdhw(x, 16).reshape(B * 3, 128, 1, H, Wc))                            # depth groups of 8 -> channels kd * 16 + c
            x = e[1](x)
            x = to_cl(to_ncdhw(x, 32).reshape(B, 96, 1, H, Wc))
            x = e[3](e[2](x))
            volume[:, :, i, :, i:] = to_ncdhw(x, self.num_features)[:, :, 0]
        return volume
