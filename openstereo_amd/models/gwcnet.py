"""GwcNet on the gfx950 cost-volume engine.

Module tree and parameter names reproduce the reference's state_dict layout
(`Backbone.feature_extraction.*`, `DispProcessor.dres0.0.0.weight`, ... -- stereo/modeling/models/
gwcnet/{gwcnet,gwcnet_backbone,gwcnet_cost_processor,gwcnet_disp_processor,hourglass}.py) so that
OpenStereo checkpoints load unchanged; the forward pass of the cost-volume / aggregation /
regression stages runs on the engine:

  features --build_cost_volume_cl--> NDHWC volume --PackedConv3d chain (MFMA)--> cost3
           --upsample_softargmin--> disparity [B,H,W]

The 2-D feature extractor runs on the same conv kernel (D = 1, NHWC, fused conv+BN+ReLU+residual launches,
SURVEY 8f #4); the PyTorch-ROCm module path stays selectable (`Backbone.use_engine = False`).
Packed weights are cached per module and rebuilt automatically when a parameter or BN statistic changes.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

import torch.nn.functional as F

from .. import autograd as AG
from .. import amp, ops, timing
from ..engine import is_split, chains, cached_pack, PackedConv3d, SmallCoConv3d, ACT_NONE, ACT_RELU


def run_train(mod, x):
    """Training-mode execution of a reference-shaped module tree: every Conv3d / ConvTranspose3d runs on
    the engine through its autograd Function; BatchNorm (batch statistics), ReLU etc. stay torch modules."""
    if isinstance(mod, (nn.Conv3d, nn.ConvTranspose3d)):
        return AG.conv_module(mod, x)
    if isinstance(mod, nn.Sequential):
        for child in mod:
            x = run_train(child, x)
        return x
    if isinstance(mod, nn.modules.batchnorm._BatchNorm):
        return AG.bn_module(mod, x)          # r6: frozen statistics (any _BatchNorm in eval mode, SyncBatchNorm included: the same affine map) on the engine; batch statistics per OSA_TRAIN_BN (never SyncBatchNorm)
    return mod(x)


# ----------------------------------------------------------------------------- 2-D backbone
def _cb2(cin, cout, k, stride, pad, dil):
    return nn.Sequential(
        nn.Conv2d(cin, cout, k, stride, dil if dil > 1 else pad, dil, bias=False),
        nn.BatchNorm2d(cout))


TRAIN_BACKBONE_ENGINE = os.environ.get("OSA_GWC_TRAIN_BACKBONE_ENGINE", "0") == "1"     # A/B switch (r5): the 2-D extractor's training path through autograd.engine_convs()
_FUSE_REDIR = os.environ.get("OSA_FUSE_REDIR", "1") != "0"
_SPLIT_ACT = os.environ.get("OSA_SPLIT_ACT", "1") != "0"     # f16x3: 3-D activations stored pre-split between engine layers
_VOL_SPLIT = os.environ.get("OSA_VOL_SPLIT", "1") != "0"     # f16x3: the cost volume itself is written in the split format (r4; A/B switch)

class _ResBlock(nn.Module):
    def __init__(self, cin, cout, stride, shortcut, pad, dil):
        super().__init__()
        self.conv1 = nn.Sequential(_cb2(cin, cout, 3, stride, pad, dil), nn.ReLU(inplace=True))
        self.conv2 = _cb2(cout, cout, 3, 1, pad, dil)
        self.downsample = shortcut

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + (x if self.downsample is None else self.downsample(x))


class _Features(nn.Module):
    """gwcnet_backbone.py:38-91: 320-channel gwc feature (+ 12-channel concat feature) at 1/4 res."""

    def __init__(self, concat_feature=True, concat_feature_channel=12):
        super().__init__()
        self.concat_feature = concat_feature
        self.firstconv = nn.Sequential(
            _cb2(3, 32, 3, 2, 1, 1), nn.ReLU(inplace=True),
            _cb2(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True),
            _cb2(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))
        self._cin = 32
        self.layer1 = self._stage(32, 3, 1, 1, 1)
        self.layer2 = self._stage(64, 16, 2, 1, 1)
        self.layer3 = self._stage(128, 3, 1, 1, 1)
        self.layer4 = self._stage(128, 3, 1, 1, 2)
        if concat_feature:
            self.lastconv = nn.Sequential(
                _cb2(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                nn.Conv2d(128, concat_feature_channel, 1, 1, 0, bias=False))

    def _stage(self, cout, n, stride, pad, dil):
        shortcut = None
        if stride != 1 or self._cin != cout:
            shortcut = nn.Sequential(nn.Conv2d(self._cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        blocks = [_ResBlock(self._cin, cout, stride, shortcut, pad, dil)]
        self._cin = cout
        blocks += [_ResBlock(cout, cout, 1, None, pad, dil) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    # ---- engine path: every conv+BN(+ReLU)(+residual) is one fused MFMA launch, NHWC activations,
    # the l2|l3|l4 concat is written in place (channel slices of one 320-channel buffer).
    def _pack(self):
        def build():
            P = lambda cb, act: PackedConv3d(cb[0], cb[1], act)
            fc = self.firstconv
            pk = {"first": [P(fc[0], ACT_RELU), P(fc[2], ACT_RELU), P(fc[4], ACT_RELU)], "layers": []}
            for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
                pk["layers"].append([(P(b.conv1[0], ACT_RELU), P(b.conv2, ACT_NONE),
                                      None if b.downsample is None else P(b.downsample, ACT_NONE)) for b in layer])
            if self.concat_feature:
                pk["last"] = (P(self.lastconv[0], ACT_RELU), PackedConv3d(self.lastconv[2]))
            return pk
        return cached_pack(self, "_pk", build)      # repacked automatically when a weight / BN statistic changes

    def forward_cl(self, img):
        """img: [N,3,H,W] (NCHW, any float dtype).  Returns (gwc_feature [N,320,1,H/4,W/4] NDHWC,
        concat_feature [N,12,1,H/4,W/4] NDHWC or None)."""
        pk = self._pack()
        # f16x3: maps that only engine layers read travel in the split hi/lo format (engine.OUT_SPLIT); the
        # l2|l3|l4 buffer the volume builder reads, and anything added to a slice of it, stay plain fp32
        sp = _SPLIT_ACT and chains(pk["first"][0].precision)
        x = ops.to_cl(img.unsqueeze(2))                 # [N,4,1,H,W], 4th channel zero
        for conv in pk["first"]:
            x = conv(x, out_split=sp)
        xoff, cat, slices = 0, None, {1: 0, 2: 64, 3: 192}
        for li, blocks in enumerate(pk["layers"]):
            for bi, (c1, c2, ds) in enumerate(blocks):
                last = bi == len(blocks) - 1
                y = c1(x, x_off=xoff, out_split=sp)
                if ds is not None:
                    skip, soff = ds(x, x_off=xoff, out_split=sp), 0
                else:
                    skip, soff = x, xoff
                if last and li >= 1:
                    if cat is None:
                        N_, _, _, h4, w4 = y.shape
                        cat = ops.empty_cl(N_, 320, 1, h4, w4, y.device)
                    c2(y, residual=skip, res_off=soff, out=cat, out_off=slices[li])
                    x, xoff = cat, slices[li]
                else:
                    x, xoff = c2(y, residual=skip, res_off=soff, out_split=sp and is_split(skip)), 0
        if not self.concat_feature:
            return cat, None
        l0, l2 = pk["last"]
        return cat, l2(l0(cat, out_split=sp))

    def forward(self, x):
        x = self.layer1(self.firstconv(x))
        l2 = self.layer2(x)
        l3 = self.layer3(l2)
        l4 = self.layer4(l3)
        out = {"gwc_feature": torch.cat((l2, l3, l4), dim=1)}
        if self.concat_feature:
            out["concat_feature"] = self.lastconv(out["gwc_feature"])
        return out


class GwcBackbone(nn.Module):
    use_engine = True        # False: the feature extractor runs as PyTorch-ROCm (MIOpen) modules

    def __init__(self, use_concat_volume=True, concat_channels=12):
        super().__init__()
        self.use_concat_volume = use_concat_volume
        self.concat_channels = concat_channels if use_concat_volume else 0
        self.feature_extraction = _Features(use_concat_volume, self.concat_channels)

    @amp.contract("cast")
    def forward(self, inputs):
        """Reference contract: NCHW feature dicts.  engine=True (default on GPU in eval mode) runs the
        extractor on the engine's conv kernel and converts at the boundary; GwcNet.forward skips
        that conversion and hands the NHWC maps straight to the volume builder."""
        left, right = inputs["left"], inputs["right"]
        B = left.shape[0]
        if self.use_engine and not self.training and ops.on_engine(left):
            gwc, catf = self.forward_cl(left, right)
            f = {"gwc_feature": ops.to_ncdhw(gwc)[:, :, 0]}
            if catf is not None:
                f["concat_feature"] = ops.to_ncdhw(catf, self.concat_channels)[:, :, 0]
        elif self.training:
            # gwcnet_backbone.py:108-109: two separate calls -- with FREEZE_BN off (the GwcNet / PSMNet default) each call has
            # its own batch statistics and its own momentum update of the running statistics
            with timing.span("backbone2d", left.shape[2], left.shape[3]):
                if TRAIN_BACKBONE_ENGINE and ops.on_engine(left):       # stride-1 convolutions forward + backward on the engine; BN / ReLU / strided convs torch
                    with AG.engine_convs():
                        return {"ref_feature": self.feature_extraction(left), "tgt_feature": self.feature_extraction(right)}
                return {"ref_feature": self.feature_extraction(left), "tgt_feature": self.feature_extraction(right)}
        else:
            with timing.span("backbone2d", left.shape[2], left.shape[3]):
                f = self.feature_extraction(torch.cat((left, right), 0))       # eval: per-sample arithmetic, one pass
        ref = {k: v[:B] for k, v in f.items()}
        tgt = {k: v[B:] for k, v in f.items()}
        return {"ref_feature": ref, "tgt_feature": tgt}

    def forward_cl(self, left, right):
        with timing.span("backbone2d_engine", left.shape[2], left.shape[3]):
            return self.feature_extraction.forward_cl(torch.cat((left, right), 0))


# ----------------------------------------------------------------------------- cost volume
class GwcVolumeCostProcessor(nn.Module):
    """gwcnet_cost_processor.py: same constructor / method names; volumes come from the engine."""

    def __init__(self, maxdisp=192, downsample=4, num_groups=40, use_concat_volume=True, *args, **kwargs):
        super().__init__()
        self.maxdisp, self.downsample = maxdisp, downsample
        self.num_groups, self.use_concat_volume = num_groups, use_concat_volume

    def build_gwc_volume(self, refimg_fea, targetimg_fea):
        return ops.build_gwc_volume(refimg_fea, targetimg_fea, self.maxdisp // self.downsample, self.num_groups)

    def build_concat_volume(self, refimg_fea, targetimg_fea):
        return ops.build_concat_volume(refimg_fea, targetimg_fea, self.maxdisp // self.downsample)

    @amp.contract("volume")
    def forward(self, inputs):
        l, r = inputs["ref_feature"], inputs["tgt_feature"]
        cat = self.use_concat_volume
        if self.training or (torch.is_grad_enabled() and l["gwc_feature"].requires_grad):
            D4 = self.maxdisp // self.downsample
            vol = AG.build_gwc_volume(l["gwc_feature"], r["gwc_feature"], D4, self.num_groups)
            if cat:
                vol = torch.cat((vol, AG.build_concat_volume(l["concat_feature"], r["concat_feature"], D4)), 1)
            return {"cost_volume": vol}
        vol = ops.build_cost_volume_cl(
            l["gwc_feature"], r["gwc_feature"], self.num_groups,
            l["concat_feature"] if cat else None, r["concat_feature"] if cat else None,
            maxdisp=self.maxdisp // self.downsample)
        return {"cost_volume": vol}

    def input_output(self):
        return {"inputs": ["ref_feature", "tgt_feature"], "outputs": ["cost_volume"]}


# ----------------------------------------------------------------------------- 3-D aggregation
def _cb3(cin, cout, k, stride, pad):
    return nn.Sequential(nn.Conv3d(cin, cout, k, stride, pad, bias=False), nn.BatchNorm3d(cout))


class Hourglass(nn.Module):
    """models/gwcnet/hourglass.py:19-56 (same parameter names); forward on the engine."""

    def __init__(self, in_channels):
        super().__init__()
        c = in_channels
        self.conv1 = nn.Sequential(_cb3(c, 2 * c, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(_cb3(2 * c, 2 * c, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv3 = nn.Sequential(_cb3(2 * c, 4 * c, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(_cb3(4 * c, 4 * c, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv5 = nn.Sequential(
            nn.ConvTranspose3d(4 * c, 2 * c, 3, padding=1, output_padding=1, stride=2, bias=False),
            nn.BatchNorm3d(2 * c))
        self.conv6 = nn.Sequential(
            nn.ConvTranspose3d(2 * c, c, 3, padding=1, output_padding=1, stride=2, bias=False),
            nn.BatchNorm3d(c))
        self.redir1 = _cb3(c, c, 1, 1, 0)
        self.redir2 = _cb3(2 * c, 2 * c, 1, 1, 0)
        self._packed = None

    def _pack(self):
        P = PackedConv3d
        return cached_pack(self, "_packed", lambda: dict(
            c1=P(self.conv1[0][0], self.conv1[0][1], ACT_RELU), c2=P(self.conv2[0][0], self.conv2[0][1], ACT_RELU),
            c3=P(self.conv3[0][0], self.conv3[0][1], ACT_RELU), c4=P(self.conv4[0][0], self.conv4[0][1], ACT_RELU),
            c5=P(self.conv5[0], self.conv5[1], ACT_RELU), c6=P(self.conv6[0], self.conv6[1], ACT_RELU),
            r1=P(self.redir1[0], self.redir1[1], ACT_NONE), r2=P(self.redir2[0], self.redir2[1], ACT_NONE)))

    def forward_cl(self, x, split=False):
        """split=True (f16x3 mode, inside GwcDispProcessor): intermediate and output tensors are written in the
        split hi/lo format by the producing epilogues (engine.OUT_SPLIT), x may be a split tensor."""
        p = self._pack()
        s = dict(out_split=True) if split else {}
        c1 = p["c1"](x, **s)
        c2 = p["c2"](c1, **s)
        c4 = p["c4"](p["c3"](c2, **s), **s)
        fuse = _FUSE_REDIR and p["c5"].precision != "f16"       # (the f16 mode has no fused redir branch: separate 1x1x1 launch)
        if fuse and p["r2"].Ci <= 64:
            c5 = p["c5"](c4, redir=(p["r2"], c2), **s)  # relu(conv5(c4) + redir2(c2)), redir2 inside conv5's epilogue
        else:
            c5 = p["c5"](c4, residual=p["r2"](c2, **s), **s)
        if fuse and p["r1"].Ci <= 32:
            return p["c6"](c5, redir=(p["r1"], x), **s)  # relu(conv6(c5) + redir1(x)), redir1 inside conv6's epilogue
        return p["c6"](c5, residual=p["r1"](x, **s), **s)    # relu(conv6(c5) + redir1(x))

    def forward_train(self, x):
        """hourglass.py:46-56 with autograd: convs on the engine, BN/ReLU/add in torch."""
        c1 = run_train(self.conv1, x)
        c2 = run_train(self.conv2, c1)
        c4 = run_train(self.conv4, run_train(self.conv3, c2))
        c5 = F.relu(run_train(self.conv5, c4) + run_train(self.redir2, c2))
        return F.relu(run_train(self.conv6, c5) + run_train(self.redir1, x))

    @amp.contract("cast")
    def forward(self, x):
        """Drop-in: NCDHW in -> NCDHW out.  Gradients required or training mode -> autograd path."""
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            return self.forward_train(x)
        return ops.to_ncdhw(self.forward_cl(ops.to_cl(x)), channels=x.shape[1])


STAGE_STASH = None          # a list: GwcDispProcessor.aggregate_cl / GwcNet.forward append (stage name, clone) of every stage (diagnostics only)


class GwcDispProcessor(nn.Module):
    """gwcnet_disp_processor.py:29-146, inference branch, on the engine."""

    def __init__(self, maxdisp=192, downsample=4, num_groups=40, use_concat_volume=True, concat_channels=12,
                 *args, **kwargs):
        super().__init__()
        self.maxdisp, self.downsample, self.num_groups = maxdisp, downsample, num_groups
        self.use_concat_volume = use_concat_volume
        self.concat_channels = concat_channels if use_concat_volume else 0
        cin = self.num_groups + self.concat_channels * 2
        relu = lambda: nn.ReLU(inplace=True)
        self.dres0 = nn.Sequential(_cb3(cin, 32, 3, 1, 1), relu(), _cb3(32, 32, 3, 1, 1), relu())
        self.dres1 = nn.Sequential(_cb3(32, 32, 3, 1, 1), relu(), _cb3(32, 32, 3, 1, 1))
        self.dres2, self.dres3, self.dres4 = Hourglass(32), Hourglass(32), Hourglass(32)
        for i in range(4):
            setattr(self, f"classif{i}", nn.Sequential(
                _cb3(32, 32, 3, 1, 1), relu(), nn.Conv3d(32, 1, kernel_size=3, padding=1, stride=1, bias=False)))
        self._packed = None

    def reset_engine(self):
        """Drop packed weights (call after loading a checkpoint or changing parameters)."""
        self._packed = None
        for h in (self.dres2, self.dres3, self.dres4):
            h._packed = None

    def _pack(self):
        P = PackedConv3d
        return cached_pack(self, "_packed", lambda: dict(
            d00=P(self.dres0[0][0], self.dres0[0][1], ACT_RELU), d02=P(self.dres0[2][0], self.dres0[2][1], ACT_RELU),
            d10=P(self.dres1[0][0], self.dres1[0][1], ACT_RELU), d12=P(self.dres1[2][0], self.dres1[2][1], ACT_NONE),
            k0=P(self.classif3[0][0], self.classif3[0][1], ACT_RELU), k2=SmallCoConv3d(self.classif3[2])),
            mods=(self.dres0, self.dres1, self.classif3))

    def aggregate_cl(self, volume):
        """NDHWC volume -> low-res cost [B,1,D/4,H/4,W/4] (classif3 output)."""
        p = self._pack()
        split = _SPLIT_ACT and chains(p["d00"].precision)
        s = dict(out_split=True) if split else {}
        if STAGE_STASH is not None:        # diagnostics (tools/diag_timed_config.py --stages): bit copies of every stage of this call
            k = lambda name, t: (STAGE_STASH.append((name, t.clone())), t)[1]
            d00 = k("dres0.0", p["d00"](k("volume", volume), **s))
            cost0 = k("dres0.2", p["d02"](d00, **s))
            cost0 = k("dres1", p["d12"](k("dres1.0", p["d10"](cost0, **s)), residual=cost0, **s))
            out1 = k("hourglass1", self.dres2.forward_cl(cost0, split))
            out2 = k("hourglass2", self.dres3.forward_cl(out1, split))
            out3 = k("hourglass3", self.dres4.forward_cl(out2, split))
            return k("classif3.2", p["k2"](k("classif3.0", p["k0"](out3))))
        cost0 = p["d02"](p["d00"](volume, **s), **s)
        cost0 = p["d12"](p["d10"](cost0, **s), residual=cost0, **s)      # dres1(cost0) + cost0
        out3 = self.dres4.forward_cl(self.dres3.forward_cl(self.dres2.forward_cl(cost0, split), split), split)
        return p["k2"](p["k0"](out3))                                   # k0 writes plain fp32 for the VALU head

    def forward_train(self, inputs):
        """gwcnet_disp_processor.py:83-126: four supervised outputs, everything differentiable."""
        volume = inputs["cost_volume"]
        h, w = inputs["left"].shape[2:]
        cost0 = run_train(self.dres0, volume)
        cost0 = run_train(self.dres1, cost0) + cost0
        out1 = self.dres2.forward_train(cost0)
        out2 = self.dres3.forward_train(out1)
        out3 = self.dres4.forward_train(out2)
        preds = []
        for head, feat in ((self.classif0, cost0), (self.classif1, out1), (self.classif2, out2), (self.classif3, out3)):
            cost = run_train(head, feat)                                  # [B,1,D/4,H/4,W/4]
            preds.append(AG.upsample_softargmin(cost, self.maxdisp, h, w, align_corners=False))
        return {"training_disp": {"disp": {"disp_ests": preds}}}

    def forward(self, inputs):
        if self.training:
            return self.forward_train(inputs)
        volume = inputs["cost_volume"]
        h, w = inputs["left"].shape[2:]
        if not ops.is_cl(volume) or volume.shape[1] % 4:
            volume = ops.to_cl(volume)
        cost3 = self.aggregate_cl(volume)
        pred3 = ops.upsample_softargmin(cost3, self.maxdisp, h, w, align_corners=False)
        return {"inference_disp": {"disp_est": pred3}}

    def input_output(self):
        return {"inputs": ["cost_volume", "disp_shape"],
                "outputs": ["training_disp", "inference_disp", "visual_summary"]}


# ----------------------------------------------------------------------------- model
class _Cfg(dict):
    __getattr__ = dict.__getitem__


GWCNET_G_SCENEFLOW = _Cfg(MAX_DISP=192, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=12, DOWNSAMPLE=4, NUM_GROUPS=40)


class GwcNet(nn.Module):
    """models/gwcnet/gwcnet.py:11-39: forward(dict{'left','right'}) -> {'disp_pred': [B,H,W]}."""

    def __init__(self, cfgs=GWCNET_G_SCENEFLOW):
        super().__init__()
        self.maxdisp = cfgs.MAX_DISP
        kw = dict(maxdisp=self.maxdisp, downsample=cfgs.DOWNSAMPLE, num_groups=cfgs.NUM_GROUPS,
                  use_concat_volume=cfgs.USE_CONCAT_VOLUME)
        self.Backbone = GwcBackbone(use_concat_volume=cfgs.USE_CONCAT_VOLUME, concat_channels=cfgs.CONCAT_CHANNELS)
        self.CostProcessor = GwcVolumeCostProcessor(**kw)
        self.DispProcessor = GwcDispProcessor(concat_channels=cfgs.CONCAT_CHANNELS, **kw)

    def reset_engine(self):
        """Drop every packed weight (call after load_state_dict / parameter updates)."""
        self.Backbone.feature_extraction._pk = None
        self.DispProcessor.reset_engine()

    def forward(self, inputs):
        if self.Backbone.use_engine and not self.training and ops.on_engine(inputs["left"]):
            # fused engine path: NHWC features never leave the engine layout
            B = inputs["left"].shape[0]
            gwc, catf = self.Backbone.forward_cl(inputs["left"], inputs["right"])
            if STAGE_STASH is not None:
                STAGE_STASH.append(("backbone gwc", gwc.clone()))
                if catf is not None:
                    STAGE_STASH.append(("backbone concat", catf.clone()))
            cp = self.CostProcessor
            # f16x3 chains: the volume is written in the chain's split format, so dres0 stages it like every later layer (ops docstring)
            vol = ops.build_cost_volume_from_cl(gwc, cp.num_groups, catf if cp.use_concat_volume else None, B,
                                                cp.maxdisp // cp.downsample,
                                                cat_channels=self.Backbone.concat_channels or None,
                                                out_split=_SPLIT_ACT and _VOL_SPLIT and self.DispProcessor._pack()["d00"].precision == "f16x3")
            if getattr(vol, "_osa_split", False):
                # the split (hi | lo fp16 halves in fp32 storage) volume is an engine-chain format: it never leaves this function.  The
                # reference publishes a real fp32 volume under "cost_volume" (inputs.update(cost_out), gwcnet.py:33-35); a tensor that
                # reports float32 but holds split halves would be silently misread by any other consumer, so the key stays
                # absent in this mode (OSA_VOL_SPLIT=0 or the exact-f32 mode publish the fp32 NDHWC volume).
                dp = self.DispProcessor
                h, w = inputs["left"].shape[2:]
                return {"disp_pred": ops.upsample_softargmin(dp.aggregate_cl(vol), dp.maxdisp, h, w, align_corners=False)}
            inputs["cost_volume"] = vol
        else:
            inputs.update(self.Backbone(inputs))
            inputs.update(self.CostProcessor(inputs))
        disp_out = self.DispProcessor(inputs)
        if self.training:
            ests = disp_out["training_disp"]["disp"]["disp_ests"]
            return {"disp_preds": ests, "disp_pred": ests[-1]}
        return {"disp_pred": disp_out["inference_disp"]["disp_est"]}

    def get_loss(self, model_preds, input_data, static=False):
        """models/gwcnet/gwcnet.py:42-53.  static=True: the same loss with static shapes and no `.item()` (masked mean written as
        sum(x * mask) / count instead of the boolean-mask gather, whose output size is a host synchronisation), so that a whole
        training step can be replayed as a hipGraph; the info dict then holds the loss tensor."""
        disp_gt = input_data["disp"]
        mask = (disp_gt < self.maxdisp) & (disp_gt > 0)
        loss = 0.0
        for disp_est, weight in zip(model_preds["disp_preds"], [0.5, 0.5, 0.7, 1.0]):
            if static:
                m = mask.to(disp_est.dtype)
                loss = loss + weight * (F.smooth_l1_loss(disp_est, disp_gt, reduction="none") * m).sum() / m.sum()
            else:
                loss = loss + weight * F.smooth_l1_loss(disp_est[mask], disp_gt[mask], reduction="mean")
        return loss, {"scalar/train/loss_disp": loss.detach() if static else loss.item()}
