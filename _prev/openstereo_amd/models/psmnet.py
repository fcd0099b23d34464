"""PSMNet on the gfx950 cost-volume engine (BASELINE configs[0]).

Parameter names reproduce the reference's state_dict (stereo/modeling/models/psmnet/
{psmnet,psmnet_backbone,psmnet_cost_processor,psmnet_disp_processor,submodule}.py) so checkpoints
load unchanged.  On the engine: cat_fms concat volume (NDHWC), the stacked-hourglass aggregator with
its cross-hourglass skips (pre/post) and cost accumulation (cost2 = classif2(out2) + cost1), and the
three trilinear(align_corners=True)+softmax+regression heads, fused.  The SPP 2-D backbone runs as
ordinary PyTorch-ROCm modules.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import autograd as AG
from .. import amp, ops, timing
from ..engine import cached_pack, PackedConv3d, SmallCoConv3d, ACT_NONE, ACT_RELU, fold_amax


# ----------------------------------------------------------------------------- 2-D backbone (submodule.py factories)
def _pad(p, d):
    return d if d > 1 else p


def _conv_bn(cin, cout, k, s, p, d, bias=True, relu=False):
    layers = [nn.Conv2d(cin, cout, k, s, _pad(p, d), d, bias=bias), nn.BatchNorm2d(cout)]
    if relu:
        layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class _Block(nn.Module):
    def __init__(self, cin, cout, stride, downsample, pad, dil):
        super().__init__()
        self.conv1 = _conv_bn(cin, cout, 3, stride, pad, dil, bias=False, relu=True)
        self.conv2 = _conv_bn(cout, cout, 3, 1, pad, dil, bias=False)
        self.downsample = downsample

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + (x if self.downsample is None else self.downsample(x))


class PSMBackbone(nn.Module):
    """psmnet_backbone.py:7-133 (SPP feature extractor -> 32 channels at 1/4 res)."""

    def __init__(self, in_planes=3):
        super().__init__()
        self.firstconv = nn.Sequential(
            _conv_bn(in_planes, 32, 3, 2, 1, 1, False, True), _conv_bn(32, 32, 3, 1, 1, 1, False, True),
            _conv_bn(32, 32, 3, 1, 1, 1, False, True))
        self._cin = 32
        self.layer1 = self._stage(32, 3, 1, 1, 1)
        self.layer2 = self._stage(64, 16, 2, 1, 1)
        self.layer3 = self._stage(128, 3, 1, 1, 1)
        self.layer4 = self._stage(128, 3, 1, 2, 2)
        for i, k in zip((1, 2, 3, 4), (64, 32, 16, 8)):
            setattr(self, f"branch{i}", nn.Sequential(nn.AvgPool2d((k, k), stride=(k, k)),
                                                      _conv_bn(128, 32, 1, 1, 0, 1, False, True)))
        self.lastconv = nn.Sequential(_conv_bn(320, 128, 3, 1, 1, 1, False, True),
                                      nn.Conv2d(128, 32, kernel_size=1, padding=0, stride=1, dilation=1, bias=False))

    def _stage(self, cout, n, stride, pad, dil):
        ds = None
        if stride != 1 or self._cin != cout:
            ds = _conv_bn(self._cin, cout, 1, stride, 0, 1)           # bias=True by default (submodule.py:32)
        blocks = [_Block(self._cin, cout, stride, ds, pad, dil)]
        self._cin = cout
        blocks += [_Block(cout, cout, 1, None, pad, dil) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def _forward(self, x):
        o2 = self.layer1(self.firstconv(x))
        o4 = self.layer2(o2)
        o8 = self.layer4(self.layer3(o4))
        size = (o8.size(2), o8.size(3))
        br = [F.interpolate(getattr(self, f"branch{i}")(o8), size, mode="bilinear", align_corners=True)
              for i in (1, 2, 3, 4)]
        return self.lastconv(torch.cat((o4, o8, br[3], br[2], br[1], br[0]), 1))

    # ---- engine path: every conv+BN(+ReLU)(+residual) is one fused MFMA launch on NHWC
    # maps; layer2 / layer4 write their outputs straight into the 320-channel SPP concat buffer
    # [o4 | o8 | branch4 | branch3 | branch2 | branch1]; only the four average pools and the bilinear
    # resampling of the 32-channel branch maps (a few hundred pixels each) stay PyTorch-ROCm ops.
    use_engine = True

    def reset_engine(self):
        self._pk = None

    def _pack(self):
        def build():
            P = lambda cb, act: PackedConv3d(cb[0], cb[1], act)
            pk = {"first": [P(m, ACT_RELU) for m in self.firstconv], "layers": []}
            for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
                pk["layers"].append([(P(b.conv1, ACT_RELU), P(b.conv2, ACT_NONE),
                                      None if b.downsample is None else P(b.downsample, ACT_NONE)) for b in layer])
            pk["branch"] = [P(getattr(self, f"branch{i}")[1], ACT_RELU) for i in (1, 2, 3, 4)]
            pk["last"] = (P(self.lastconv[0], ACT_RELU), PackedConv3d(self.lastconv[1]))
            return pk
        return cached_pack(self, "_pk", build)

    def forward_cl(self, img):
        """img [N,3,H,W] -> NHWC feature map (logical [N,32,1,H/4,W/4])."""
        pk = self._pack()
        x = ops.to_cl(img.unsqueeze(2))                 # [N,4,1,H,W], 4th channel zero
        for conv in pk["first"]:
            x = conv(x)
        xoff, cat = 0, None
        slices = {1: 0, 3: 64}                          # layer2 -> cat[0:64] (o4), layer4 -> cat[64:192] (o8)
        for li, blocks in enumerate(pk["layers"]):
            for bi, (c1, c2, ds) in enumerate(blocks):
                y = c1(x, x_off=xoff)
                skip, soff = (ds(x, x_off=xoff), 0) if ds is not None else (x, xoff)
                if bi == len(blocks) - 1 and li in slices:
                    if cat is None:
                        N_, _, _, h4, w4 = y.shape
                        cat = ops.empty_cl(N_, 320, 1, h4, w4, y.device)
                    c2(y, residual=skip, res_off=soff, out=cat, out_off=slices[li])
                    x, xoff = cat, slices[li]
                else:
                    x, xoff = c2(y, residual=skip, res_off=soff), 0
        h4, w4 = cat.shape[3], cat.shape[4]
        o8 = cat[:, 64:192, 0]                          # NCHW-logical view of the NHWC slice
        for slot, (i, k) in enumerate(((4, 8), (3, 16), (2, 32), (1, 64))):      # concat order branch4..branch1 (psmnet_backbone.py:127)
            pooled = F.avg_pool2d(o8, (k, k), stride=(k, k))
            N_, C_, ph, pw = pooled.shape
            pcl = ops.empty_cl(N_, C_, 1, ph, pw, pooled.device)
            pcl[:, :, 0] = pooled
            br = pk["branch"][i - 1](pcl)                                      # 1x1 conv + BN + ReLU
            up = F.interpolate(br[:, :, 0], (h4, w4), mode="bilinear", align_corners=True)
            cat[:, 192 + 32 * slot:224 + 32 * slot, 0] = up
            if getattr(cat, "_osa_meta", None) is not None:            # f16x3: torch wrote into an engine buffer
                fold_amax(cat, up)
        l0, l1 = pk["last"]
        return l1(l0(cat))

    @amp.contract("cast")
    def forward(self, inputs):
        left, right = inputs["left"], inputs["right"]
        B = left.shape[0]
        with timing.span("backbone2d", left.shape[2], left.shape[3]):
            if self.training:          # psmnet_backbone.py:119-133: separate calls (separate BatchNorm batch statistics)
                return {"ref_feature": self._forward(left), "tgt_feature": self._forward(right)}
            x = torch.cat((left, right), 0)
            if self.use_engine and ops.on_engine(x):
                f = self.forward_cl(x)[:, :32, 0].contiguous()
            else:
                f = self._forward(x)
        return {"ref_feature": f[:B], "tgt_feature": f[B:]}


# ----------------------------------------------------------------------------- 3-D aggregator
def _c3(cin, cout, k, s, p, relu):
    layers = [nn.Conv3d(cin, cout, k, s, p, bias=False), nn.BatchNorm3d(cout)]
    if relu:
        layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


def _d3(cin, cout):
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False),
                         nn.BatchNorm3d(cout))


class Hourglass(nn.Module):
    """psmnet_cost_processor.py:53-132: forward(x, presqu, postsqu) -> (out, pre, post)."""

    def __init__(self, in_planes, batch_norm=True):
        super().__init__()
        assert batch_norm
        c = in_planes
        self.conv1 = _c3(c, 2 * c, 3, 2, 1, True)
        self.conv2 = _c3(2 * c, 2 * c, 3, 1, 1, False)
        self.conv3 = _c3(2 * c, 2 * c, 3, 2, 1, True)
        self.conv4 = _c3(2 * c, 2 * c, 3, 1, 1, True)
        self.conv5 = _d3(2 * c, 2 * c)
        self.conv6 = _d3(2 * c, c)
        self._packed = None

    def _pack(self):
        P = PackedConv3d
        return cached_pack(self, "_packed", lambda: dict(
            c1=P(self.conv1[0], self.conv1[1], ACT_RELU), c2=P(self.conv2[0], self.conv2[1], ACT_RELU),
            c3=P(self.conv3[0], self.conv3[1], ACT_RELU), c4=P(self.conv4[0], self.conv4[1], ACT_RELU),
            c5=P(self.conv5[0], self.conv5[1], ACT_RELU), c6=P(self.conv6[0], self.conv6[1], ACT_NONE)))

    def forward_cl(self, x, presqu=None, postsqu=None, out_residual=None):
        """NDHWC tensors. out_residual is added to conv6's output (the aggregator's `out + cost0`)."""
        p = self._pack()
        out = p["c1"](x)
        pre = p["c2"](out, residual=postsqu)                   # relu(conv2(out) [+ postsqu])
        out = p["c4"](p["c3"](pre))
        post = p["c5"](out, residual=presqu if presqu is not None else pre)   # relu(conv5(out) + presqu|pre)
        return p["c6"](post, residual=out_residual), pre, post

    def forward_train(self, x, presqu=None, postsqu=None):
        """psmnet_cost_processor.py:108-132 as a torch composition; under AG.engine_convs() every Conv3d / ConvTranspose3d (forward,
        dgrad, wgrad) runs on the engine, BatchNorm (batch statistics) / ReLU / adds are the reference's torch ops."""
        with AG.engine_convs():
            out = self.conv1(x)
            pre = self.conv2(out)
            pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)
            out = self.conv4(self.conv3(pre))
            post = F.relu(self.conv5(out) + (presqu if presqu is not None else pre))
            return self.conv6(post), pre, post

    @amp.contract("cast")
    def forward(self, x, presqu=None, postsqu=None):
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            return self.forward_train(x, presqu, postsqu)
        cl = lambda t: None if t is None else ops.to_cl(t)
        out, pre, post = self.forward_cl(ops.to_cl(x), cl(presqu), cl(postsqu))
        return ops.to_ncdhw(out), ops.to_ncdhw(pre), ops.to_ncdhw(post)


class PSMAggregator(nn.Module):
    """psmnet_cost_processor.py:135-221.  forward returns the reference's [cost3, cost2, cost1]
    (full-resolution costs, drop-in); aggregate_cl returns the low-res costs the fused heads use."""

    def __init__(self, max_disp, in_planes=64, batch_norm=True):
        super().__init__()
        self.max_disp, self.in_planes = max_disp, in_planes
        self.dres0 = nn.Sequential(_c3(in_planes, 32, 3, 1, 1, True), _c3(32, 32, 3, 1, 1, True))
        self.dres1 = nn.Sequential(_c3(32, 32, 3, 1, 1, True), _c3(32, 32, 3, 1, 1, False))
        self.dres2, self.dres3, self.dres4 = Hourglass(32), Hourglass(32), Hourglass(32)
        for i in (1, 2, 3):
            setattr(self, f"classif{i}", nn.Sequential(_c3(32, 32, 3, 1, 1, True),
                                                       nn.Conv3d(32, 1, kernel_size=3, stride=1, padding=1, bias=False)))
        self._packed = None

    def reset_engine(self):
        self._packed = None
        for h in (self.dres2, self.dres3, self.dres4):
            h._packed = None

    def _pack(self):
        def build():
            P = PackedConv3d
            d = dict(d00=P(self.dres0[0][0], self.dres0[0][1], ACT_RELU), d01=P(self.dres0[1][0], self.dres0[1][1], ACT_RELU),
                     d10=P(self.dres1[0][0], self.dres1[0][1], ACT_RELU), d11=P(self.dres1[1][0], self.dres1[1][1], ACT_NONE))
            for i in (1, 2, 3):
                c = getattr(self, f"classif{i}")
                d[f"k{i}a"], d[f"k{i}b"] = P(c[0][0], c[0][1], ACT_RELU), SmallCoConv3d(c[1])
            return d
        return cached_pack(self, "_packed", build, mods=(self.dres0, self.dres1, self.classif1, self.classif2, self.classif3))

    def aggregate_cl(self, raw_cost):
        p = self._pack()
        cost0 = p["d01"](p["d00"](raw_cost))
        cost0 = p["d11"](p["d10"](cost0), residual=cost0)
        out1, pre1, post1 = self.dres2.forward_cl(cost0, None, None, out_residual=cost0)
        out2, pre2, post2 = self.dres3.forward_cl(out1, pre1, post1, out_residual=cost0)
        out3, _, _ = self.dres4.forward_cl(out2, pre2, post2, out_residual=cost0)
        cost1 = p["k1b"](p["k1a"](out1))
        cost2 = p["k2b"](p["k2a"](out2), residual=cost1)
        cost3 = p["k3b"](p["k3a"](out3), residual=cost2)
        return cost3, cost2, cost1

    def aggregate_train(self, raw_cost):
        """psmnet_cost_processor.py:182-198 -> low-resolution costs (cost3, cost2, cost1), differentiable on the engine."""
        with AG.engine_convs():
            cost0 = self.dres0(raw_cost)
            cost0 = self.dres1(cost0) + cost0
            out1, pre1, post1 = self.dres2.forward_train(cost0, None, None)
            out1 = out1 + cost0
            out2, pre2, post2 = self.dres3.forward_train(out1, pre1, post1)
            out2 = out2 + cost0
            out3, _, _ = self.dres4.forward_train(out2, pre2, post2)
            out3 = out3 + cost0
            cost1 = self.classif1(out1)
            cost2 = self.classif2(out2) + cost1
            cost3 = self.classif3(out3) + cost2
        return cost3, cost2, cost1

    @amp.contract("cast")
    def forward(self, raw_cost):
        B, C, D, H, W = raw_cost.shape
        if self.training or (torch.is_grad_enabled() and raw_cost.requires_grad):
            # drop-in contract: full-resolution costs (:200-221); the engine model (PSMCostProcessor) keeps them low-res and fuses the upsample
            return [F.interpolate(c, [self.max_disp, H * 4, W * 4], mode="trilinear", align_corners=True).squeeze(1)
                    for c in self.aggregate_train(raw_cost)]
        lows = self.aggregate_cl(ops.to_cl(raw_cost))
        # drop-in contract: full-resolution costs (psmnet_cost_processor.py:200-221); torch does the upsample
        return [F.interpolate(c, [self.max_disp, H * 4, W * 4], mode="trilinear", align_corners=True).squeeze(1)
                for c in lows]


class PSMCostProcessor(nn.Module):
    def __init__(self, max_disp=192, in_planes=64):
        super().__init__()
        self.max_disp = max_disp
        self.aggregator = PSMAggregator(max_disp=max_disp, in_planes=in_planes)

    def cat_func(self, left, right):
        return ops.cat_fms(left, right, max_disp=int(self.max_disp // 4), start_disp=0, dilation=1)

    @amp.contract("cast")
    def forward(self, inputs):
        """Engine path: keeps the three costs at 1/4 resolution (the fused heads upsample on the fly)."""
        l, r = inputs["ref_feature"], inputs["tgt_feature"]
        if self.training or (torch.is_grad_enabled() and (l.requires_grad or r.requires_grad)):
            vol = AG.build_concat_volume(l.float(), r.float(), int(self.aggregator.max_disp // 4))          # cat_fms, differentiable
            cost3, cost2, cost1 = self.aggregator.aggregate_train(vol)
            return {"cost1": cost1, "cost2": cost2, "cost3": cost3}
        vol = ops.build_cost_volume_cl(None, None, 0, l, r, maxdisp=int(self.aggregator.max_disp // 4))   # attribute the reference class has too
        cost3, cost2, cost1 = self.aggregator.aggregate_cl(vol)
        return {"cost1": cost1, "cost2": cost2, "cost3": cost3}

    def input_output(self):
        return {"inputs": ["ref_feature", "tgt_feature"], "outputs": ["cost1", "cost2", "cost3"]}


class PSMDispProcessor(nn.Module):
    """psmnet_disp_processor.py:93-118; accepts the reference's full-res costs [B,D,H,W] or the
    engine's low-res costs [B,1,D/4,H/4,W/4] (then upsample+softmax+regression run fused)."""

    def __init__(self, max_disp=192):
        super().__init__()
        self.max_disp = max_disp
        self.disp_processor = ops.FasterSoftArgmin(max_disp=max_disp, start_disp=0, dilation=1, alpha=1.0, normalize=True)

    def forward(self, inputs):
        h, w = inputs["left"].shape[2:]
        out = []
        for k in ("cost1", "cost2", "cost3"):
            c = inputs[k]
            if c.dim() == 5 and torch.is_grad_enabled() and c.requires_grad:      # training: fused head with its backward kernel
                out.append(AG.upsample_softargmin(c, self.disp_processor.max_disp, h, w, align_corners=True))
            elif c.dim() == 5:
                out.append(ops.upsample_softargmin(c, self.disp_processor.max_disp, h, w, align_corners=True))
            else:
                out.append(self.disp_processor(c))
        return out


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class PSMNet(nn.Module):
    """models/psmnet/psmnet.py:11-30: forward(dict) -> {'disp_pred': disp3, 'train_preds': [disp1,disp2,disp3]}."""

    def __init__(self, cfgs=_Cfg(MAX_DISP=192)):
        super().__init__()
        self.maxdisp = cfgs.MAX_DISP
        self.Backbone = PSMBackbone()
        self.CostProcessor = PSMCostProcessor(max_disp=self.maxdisp)
        self.DispProcessor = PSMDispProcessor(max_disp=self.maxdisp)

    def forward(self, inputs):
        inputs.update(self.Backbone(inputs))
        inputs.update(self.CostProcessor(inputs))
        disp_out = self.DispProcessor(inputs)
        return {"disp_pred": disp_out[-1], "train_preds": disp_out}

    def get_loss(self, model_preds, input_data):
        """models/psmnet/psmnet.py:32-45"""
        disp_gt = input_data["disp"]
        mask = (disp_gt < self.maxdisp) & (disp_gt > 0)
        loss = 0.0
        for pred, weight in zip(model_preds["train_preds"], [0.5, 0.7, 1.0]):
            loss = loss + weight * F.smooth_l1_loss(pred[mask], disp_gt[mask], reduction="mean")
        return loss, {"scalar/train/loss_disp": loss.item()}
