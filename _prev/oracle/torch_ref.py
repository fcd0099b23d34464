"""CPU ORACLE -- test infrastructure only.  Never imported by the product path.

A restatement of OpenStereo's hot path (cost-volume build -> 3-D aggregation -> soft-argmin) as
plain functions over a flat state_dict, executed with torch *CPU* fp32 operators -- the same
arithmetic the reference's nn.Modules run on its CPU path.  Every function cites the reference
file:line it follows (paths relative to the OpenStereo tree, stereo/modeling/...).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Pinning: tests/golden/*.npz hold outputs of the REAL reference (imported from /root/reference by
tests/golden/make_golden.py in the build container); tests/test_oracle_golden.py checks this
restatement against them.  Parity status: pinned for GwcNet and PSMNet (full models + every stage), the
shared / IGEV volume + regression helpers, the StereoBase / IGEV / LightStereo cost stages and update
block, and -- since r3 -- the StereoBase / IGEVStereo / LightStereo whole models, forward and CPU
autograd, against the reference's OWN classes (tests/golden/e2e_reference*.npz).  "Parity unpinned":
only the timm feature trunks themselves (package / weights unavailable offline); the fixtures
replace exactly `model.blocks` by the same stand-in on both sides.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- volumes
def gwc_volume(left, right, maxdisp, num_groups):
    """cost_volume.py:59-78 / gwcnet_cost_processor.py:13-39 / igev/submodule.py:158-177.
    V[b,g,d,h,w] = mean_k L[b,gK+k,h,w] * R[b,gK+k,h,w-d]  (w >= d), else 0."""
    B, C, H, W = left.shape
    assert C % num_groups == 0
    K = C // num_groups
    vol = left.new_zeros(B, num_groups, maxdisp, H, W)
    for d in range(min(maxdisp, W)):
        prod = left[..., d:] * right[..., : W - d]
        vol[:, :, d, :, d:] = prod.view(B, num_groups, K, H, W - d).mean(dim=2)
    return vol


def concat_volume(left, right, maxdisp, mask_left=True):
    """cost_volume.py:81-92 / gwcnet_cost_processor.py:41-53 / psmnet_cost_processor.py:9-50
    (start_disp=0, dilation=1).  mask_left=False: igev/submodule.py:216-227 (left half unmasked)."""
    B, C, H, W = left.shape
    vol = left.new_zeros(B, 2 * C, maxdisp, H, W)
    for d in range(maxdisp):
        if mask_left:
            if d < W:
                vol[:, :C, d, :, d:] = left[..., d:]
        else:
            vol[:, :C, d] = left
        if d < W:
            vol[:, C:, d, :, d:] = right[..., : W - d]
    return vol


def corr_volume(left, right, maxdisp):
    """cost_volume.py:32-41 (correlation_volume): one group, mean over C."""
    B, C, H, W = left.shape
    vol = left.new_zeros(B, maxdisp, H, W)
    for d in range(min(maxdisp, W)):
        vol[:, d, :, d:] = (left[..., d:] * right[..., : W - d]).mean(dim=1)
    return vol


def build_corr_volume(left, right, maxdisp):
    """cost_volume.py:95-105: as correlation_volume, except that planes d >= W take the *else*
    branch of `(i > 0) & (i < W)` and therefore repeat the unshifted d=0 correlation."""
    vol = corr_volume(left, right, maxdisp)
    W = left.shape[-1]
    if maxdisp > W:
        vol[:, W:] = vol[:, :1]
    return vol


# ----------------------------------------------------------------------------- regression
def disparity_regression(prob, maxdisp, keepdim=True):
    """disp_regression.py:8-12 (keepdim=True); gwcnet_disp_processor.py:22-26 (keepdim=False)."""
    assert prob.dim() == 4
    d = torch.arange(0, maxdisp, dtype=prob.dtype, device=prob.device).view(1, maxdisp, 1, 1)
    return torch.sum(prob * d, 1, keepdim=keepdim)


def softmax_regression(cost, keepdim=True):
    """F.softmax(dim=1) + regression (stereobase_gru.py:163-164, igev_stereo.py:164-165,
    psmnet_disp_processor.py:64-71 with alpha=1, normalize=True)."""
    return disparity_regression(F.softmax(cost, dim=1), cost.shape[1], keepdim)


def upsample_regression(cost_lowres, maxdisp, h, w, align_corners=False):
    """gwcnet_disp_processor.py:128-133: trilinear upsample of [B,1,Dl,Hl,Wl] to [maxdisp,h,w],
    squeeze, softmax over D, expectation (keepdim=False).  PSMNet: align_corners=True
    (psmnet_cost_processor.py:201-214)."""
    if cost_lowres.dim() == 4:
        cost_lowres = cost_lowres[:, None]
    c = F.interpolate(cost_lowres, [maxdisp, h, w], mode="trilinear", align_corners=align_corners)
    return disparity_regression(F.softmax(c.squeeze(1), dim=1), maxdisp, keepdim=False)


# ----------------------------------------------------------------------------- conv blocks
def _bn(x, sd, p, eps=1e-5):
    """eval-mode BatchNorm (running statistics), default eps (gwcnet_disp_processor.py:8-19)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=eps)


def convbn3d(x, sd, p, stride=1, pad=1):
    """convbn_3d = Sequential(Conv3d(bias=False), BatchNorm3d): keys p.0.weight, p.1.*"""
    return _bn(F.conv3d(x, sd[p + ".0.weight"], None, stride, pad), sd, p + ".1")


def gwc_hourglass(x, sd, p):
    """models/gwcnet/hourglass.py:46-56."""
    c1 = F.relu(convbn3d(x, sd, p + ".conv1.0", 2, 1))
    c2 = F.relu(convbn3d(c1, sd, p + ".conv2.0", 1, 1))
    c3 = F.relu(convbn3d(c2, sd, p + ".conv3.0", 2, 1))
    c4 = F.relu(convbn3d(c3, sd, p + ".conv4.0", 1, 1))
    up5 = _bn(F.conv_transpose3d(c4, sd[p + ".conv5.0.weight"], None, 2, 1, 1), sd, p + ".conv5.1")
    c5 = F.relu(up5 + convbn3d(c2, sd, p + ".redir2", 1, 0))
    up6 = _bn(F.conv_transpose3d(c5, sd[p + ".conv6.0.weight"], None, 2, 1, 1), sd, p + ".conv6.1")
    return F.relu(up6 + convbn3d(x, sd, p + ".redir1", 1, 0))


def gwc_aggregate(volume, sd, p="DispProcessor", taps=None):
    """gwcnet_disp_processor.py:83-91,128-129 (inference branch): volume [B,64,D4,H4,W4] -> cost3 [B,1,D4,H4,W4].
    `taps` (dict) receives every intermediate stage tensor when given."""
    t = {} if taps is None else taps
    x = F.relu(convbn3d(volume, sd, p + ".dres0.0"))
    x = F.relu(convbn3d(x, sd, p + ".dres0.2"))
    t["dres0"] = x
    y = F.relu(convbn3d(x, sd, p + ".dres1.0"))
    cost0 = convbn3d(y, sd, p + ".dres1.2") + x
    t["cost0"] = cost0
    out1 = gwc_hourglass(cost0, sd, p + ".dres2"); t["out1"] = out1
    out2 = gwc_hourglass(out1, sd, p + ".dres3"); t["out2"] = out2
    out3 = gwc_hourglass(out2, sd, p + ".dres4"); t["out3"] = out3
    z = F.relu(convbn3d(out3, sd, p + ".classif3.0"))
    cost3 = F.conv3d(z, sd[p + ".classif3.2.weight"], None, 1, 1)
    t["cost3"] = cost3
    return cost3


# ----------------------------------------------------------------------------- GwcNet 2-D features
def _convbn2d(x, sd, p, stride, pad, dil):
    """gwcnet_backbone.py:6-10 (padding = dilation if dilation > 1 else pad)."""
    return _bn(F.conv2d(x, sd[p + ".0.weight"], None, stride, dil if dil > 1 else pad, dil), sd, p + ".1")


def _basic_block(x, sd, p, stride, pad, dil):
    """gwcnet_backbone.py:13-35 (no ReLU after the residual add)."""
    y = F.relu(_convbn2d(x, sd, p + ".conv1.0", stride, pad, dil))
    y = _convbn2d(y, sd, p + ".conv2", 1, pad, dil)
    if (p + ".downsample.0.weight") in sd:
        x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), sd, p + ".downsample.1")
    return y + x


def gwc_features(img, sd, p="Backbone.feature_extraction", concat=True):
    """gwcnet_backbone.py:78-91."""
    x = F.relu(_convbn2d(img, sd, p + ".firstconv.0", 2, 1, 1))
    x = F.relu(_convbn2d(x, sd, p + ".firstconv.2", 1, 1, 1))
    x = F.relu(_convbn2d(x, sd, p + ".firstconv.4", 1, 1, 1))
    for i in range(3):
        x = _basic_block(x, sd, f"{p}.layer1.{i}", 1, 1, 1)
    l2 = x
    for i in range(16):
        l2 = _basic_block(l2, sd, f"{p}.layer2.{i}", 2 if i == 0 else 1, 1, 1)
    l3 = l2
    for i in range(3):
        l3 = _basic_block(l3, sd, f"{p}.layer3.{i}", 1, 1, 1)
    l4 = l3
    for i in range(3):
        l4 = _basic_block(l4, sd, f"{p}.layer4.{i}", 1, 1, 2)
    gwc = torch.cat((l2, l3, l4), dim=1)
    if not concat:
        return gwc, None
    y = F.relu(_convbn2d(gwc, sd, p + ".lastconv.0", 1, 1, 1))
    return gwc, F.conv2d(y, sd[p + ".lastconv.2.weight"])


def gwcnet_forward(left, right, sd, maxdisp=192, downsample=4, num_groups=40, concat=True, taps=None):
    """models/gwcnet/gwcnet.py:27-39, inference: {'left','right'} -> disp_pred [B,H,W]."""
    t = {} if taps is None else taps
    lg, lc = gwc_features(left, sd, concat=concat)
    rg, rc = gwc_features(right, sd, concat=concat)
    t["left_gwc"], t["right_gwc"], t["left_cat"], t["right_cat"] = lg, rg, lc, rc
    D4 = maxdisp // downsample
    vol = gwc_volume(lg, rg, D4, num_groups)
    if concat:
        vol = torch.cat((vol, concat_volume(lc, rc, D4)), 1)      # gwcnet_cost_processor.py:65
    t["volume"] = vol
    cost3 = gwc_aggregate(vol, sd, taps=t)
    h, w = left.shape[2:]
    disp = upsample_regression(cost3, maxdisp, h, w, align_corners=False)
    t["disp"] = disp
    return disp


def gwc_hot_path(lg, rg, lc, rc, sd, maxdisp, h, w, num_groups=40):
    """Features -> disparity: the engine's scope (everything after the 2-D backbone)."""
    D4 = maxdisp // 4
    vol = torch.cat((gwc_volume(lg, rg, D4, num_groups), concat_volume(lc, rc, D4)), 1)
    return upsample_regression(gwc_aggregate(vol, sd), maxdisp, h, w, align_corners=False)


# ============================================================================= PSMNet
def _cbr2(x, sd, p, stride, pad, dil, relu=True):
    """submodule.py:14-43,100-117 conv_bn[_relu] (2-D): keys p.0.(weight[,bias]), p.1.*"""
    y = F.conv2d(x, sd[p + ".0.weight"], sd.get(p + ".0.bias"), stride, dil if dil > 1 else pad, dil)
    y = _bn(y, sd, p + ".1")
    return F.relu(y) if relu else y


def _psm_block(x, sd, p, stride, pad, dil):
    """submodule.py:219-245 BasicBlock (out += x, no ReLU after)."""
    y = _cbr2(x, sd, p + ".conv1", stride, pad, dil, True)
    y = _cbr2(y, sd, p + ".conv2", 1, pad, dil, False)
    if (p + ".downsample.0.weight") in sd:
        x = _cbr2(x, sd, p + ".downsample", stride, 0, 1, False)
    return y + x


def psm_features(img, sd, p="Backbone"):
    """psmnet_backbone.py:84-116 (SPP)."""
    x = img
    for i, s in enumerate((2, 1, 1)):
        x = _cbr2(x, sd, f"{p}.firstconv.{i}", s, 1, 1)
    for i in range(3):
        x = _psm_block(x, sd, f"{p}.layer1.{i}", 1, 1, 1)
    o4 = x
    for i in range(16):
        o4 = _psm_block(o4, sd, f"{p}.layer2.{i}", 2 if i == 0 else 1, 1, 1)
    o8 = o4
    for i in range(3):
        o8 = _psm_block(o8, sd, f"{p}.layer3.{i}", 1, 1, 1)
    for i in range(3):
        o8 = _psm_block(o8, sd, f"{p}.layer4.{i}", 1, 2, 2)
    size = o8.shape[2:]
    br = []
    for i, k in zip((1, 2, 3, 4), (64, 32, 16, 8)):
        b = _cbr2(F.avg_pool2d(o8, (k, k), (k, k)), sd, f"{p}.branch{i}.1", 1, 0, 1)
        br.append(F.interpolate(b, size, mode="bilinear", align_corners=True))
    f = torch.cat((o4, o8, br[3], br[2], br[1], br[0]), 1)
    f = _cbr2(f, sd, p + ".lastconv.0", 1, 1, 1)
    return F.conv2d(f, sd[p + ".lastconv.1.weight"])


def _c3(x, sd, p, stride, relu):
    y = _bn(F.conv3d(x, sd[p + ".0.weight"], None, stride, 1), sd, p + ".1")
    return F.relu(y) if relu else y


def _d3(x, sd, p):
    return _bn(F.conv_transpose3d(x, sd[p + ".0.weight"], None, 2, 1, 1), sd, p + ".1")


def psm_hourglass(x, sd, p, presqu=None, postsqu=None):
    """psmnet_cost_processor.py:108-132."""
    out = _c3(x, sd, p + ".conv1", 2, True)
    pre = _c3(out, sd, p + ".conv2", 1, False)
    pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)
    out = _c3(_c3(pre, sd, p + ".conv3", 2, True), sd, p + ".conv4", 1, True)
    post = F.relu(_d3(out, sd, p + ".conv5") + (presqu if presqu is not None else pre))
    return _d3(post, sd, p + ".conv6"), pre, post


def psm_aggregate(raw_cost, sd, p="CostProcessor.aggregator", taps=None):
    """psmnet_cost_processor.py:182-198 -> low-res (cost3, cost2, cost1), each [B,1,D/4,H/4,W/4]."""
    t = {} if taps is None else taps
    c0 = _c3(_c3(raw_cost, sd, p + ".dres0.0", 1, True), sd, p + ".dres0.1", 1, True)
    c0 = _c3(_c3(c0, sd, p + ".dres1.0", 1, True), sd, p + ".dres1.1", 1, False) + c0
    t["cost0"] = c0
    out1, pre1, post1 = psm_hourglass(c0, sd, p + ".dres2")
    out1 = out1 + c0
    out2, pre2, post2 = psm_hourglass(out1, sd, p + ".dres3", pre1, post1)
    out2 = out2 + c0
    out3, _, _ = psm_hourglass(out2, sd, p + ".dres4", pre2, post2)
    out3 = out3 + c0
    t["out1"], t["out3"] = out1, out3
    head = lambda x, q: F.conv3d(_c3(x, sd, q + ".0", 1, True), sd[q + ".1.weight"], None, 1, 1)
    cost1 = head(out1, p + ".classif1")
    cost2 = head(out2, p + ".classif2") + cost1
    cost3 = head(out3, p + ".classif3") + cost2
    t["cost1"], t["cost3"] = cost1, cost3
    return cost3, cost2, cost1


def psmnet_forward(left, right, sd, maxdisp=192, taps=None):
    """models/psmnet/psmnet.py:20-29 -> [disp1, disp2, disp3] (each [B,H,W]); trilinear align_corners=True."""
    t = {} if taps is None else taps
    lf, rf = psm_features(left, sd), psm_features(right, sd)
    t["left_feature"], t["right_feature"] = lf, rf
    vol = concat_volume(lf, rf, maxdisp // 4)
    cost3, cost2, cost1 = psm_aggregate(vol, sd, taps=t)
    h, w = left.shape[2:]
    return [upsample_regression(c, maxdisp, h, w, align_corners=True) for c in (cost1, cost2, cost3)]


# ============================================================================= StereoBase / IGEV aggregation (a8)
def _sb_names(style):
    """Parameter sub-keys of one conv unit: StereoBase BasicConv3d (`.block.0/.block.1`,
    common/basic_block_3d.py:5-38) or IGEV BasicConv (`.conv/.bn`, models/igev/submodule.py:6-32)."""
    return (".block.0", ".block.1") if style == "stereobase" else (".conv", ".bn")


def _unit3d(x, sd, p, style, stride=1, pad=1, deconv=False, bn=True, act=True):
    cw, cb = _sb_names(style)
    if deconv:
        y = F.conv_transpose3d(x, sd[p + cw + ".weight"], None, 2, 1)          # k4 s2 p1
    else:
        y = F.conv3d(x, sd[p + cw + ".weight"], None, stride, pad)
    if bn:
        y = _bn(y, sd, p + cb)
    return F.leaky_relu(y, 0.01) if act else y


def _feature_att(cv, feat, sd, p, style):
    """stereobase/igev_blocks.py:35-48 / igev/submodule.py:237-250: cv * sigmoid(Conv2d(lrelu(bn(conv1x1(feat)))))."""
    cw, cb = _sb_names(style)
    a = F.leaky_relu(_bn(F.conv2d(feat, sd[p + ".feat_att.0" + cw + ".weight"]), sd, p + ".feat_att.0" + cb), 0.01)
    a = F.conv2d(a, sd[p + ".feat_att.1.weight"], sd[p + ".feat_att.1.bias"])
    return torch.sigmoid(a.unsqueeze(2)) * cv


def igev_style_hourglass(x, features, sd, p, style="stereobase", return_multi=False):
    """models/stereobase/hourglass.py:79-104 == models/igev/igev_stereo.py:51-76."""
    u = lambda t, q, **kw: _unit3d(t, sd, p + q, style, **kw)
    conv1 = u(u(x, ".conv1.0", stride=2), ".conv1.1")
    conv1 = _feature_att(conv1, features[1], sd, p + ".feature_att_8", style)
    conv2 = u(u(conv1, ".conv2.0", stride=2), ".conv2.1")
    conv2 = _feature_att(conv2, features[2], sd, p + ".feature_att_16", style)
    conv3 = u(u(conv2, ".conv3.0", stride=2), ".conv3.1")
    conv3 = _feature_att(conv3, features[3], sd, p + ".feature_att_32", style)
    conv3_up = u(conv3, ".conv3_up", deconv=True)
    conv2 = torch.cat((conv3_up, conv2), dim=1)
    conv2 = u(u(u(conv2, ".agg_0.0", pad=0), ".agg_0.1"), ".agg_0.2")
    conv2 = _feature_att(conv2, features[2], sd, p + ".feature_att_up_16", style)
    conv2_up = u(conv2, ".conv2_up", deconv=True)
    conv1 = torch.cat((conv2_up, conv1), dim=1)
    conv1 = u(u(u(conv1, ".agg_1.0", pad=0), ".agg_1.1"), ".agg_1.2")
    conv1 = _feature_att(conv1, features[1], sd, p + ".feature_att_up_8", style)
    conv = u(conv1, ".conv1_up", deconv=True, bn=False, act=False)
    return [conv, conv1, conv2] if return_multi else conv


def stereobase_cost_stage(match_l, match_r, cat_l, cat_r, features, sd, max_disp, num_groups=8):
    """stereobase_gru.py:139-164: gwc + concat volume -> cost_agg -> classifier -> softmax -> regression."""
    D4 = max_disp // 4
    vol = torch.cat((gwc_volume(match_l, match_r, D4, num_groups), concat_volume(cat_l, cat_r, D4)), 1)
    geo = igev_style_hourglass(vol, features, sd, "cost_agg", "stereobase")
    prob = F.softmax(F.conv3d(geo, sd["classifier.weight"], None, 1, 1).squeeze(1), dim=1)
    return disparity_regression(prob, D4, keepdim=True), prob, geo


def igev_cost_stage(match_l, match_r, features, sd, max_disp):
    """igev_stereo.py:158-168: gwc(8) volume -> corr_stem -> corr_feature_att -> cost_agg -> classifier -> softmax -> regression."""
    D4 = max_disp // 4
    vol = _unit3d(gwc_volume(match_l, match_r, D4, 8), sd, "corr_stem", "igev")
    vol = _feature_att(vol, features[0], sd, "corr_feature_att", "igev")
    geo = igev_style_hourglass(vol, features, sd, "cost_agg", "igev")
    prob = F.softmax(F.conv3d(geo, sd["classifier.weight"], None, 1, 1).squeeze(1), dim=1)
    return disparity_regression(prob, D4, keepdim=True), prob, geo


# ============================================================================= dormant volume variants
def coex_cost_volume(x, y, maxdisp, group=1):
    """cost_volume.py:9-29 (CoExCostVolume.forward): cost[b,g,d,h,w] = sum_k x[g,k,h,w] * y[g,k,h,w-d], d = 0..maxdisp."""
    b, c, h, w = x.shape
    xg, yg = x.reshape(b, group, c // group, h, w), y.reshape(b, group, c // group, h, w)
    out = x.new_zeros(b, group, maxdisp + 1, h, w)
    for d in range(maxdisp + 1):
        if d < w:
            out[:, :, d, :, d:] = (xg[..., d:] * yg[..., :w - d]).sum(2)
    return out


def cat_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1):
    """psmnet_cost_processor.py:9-50: concat volume over the disparities int(linspace(start, start + max_disp - 1, n)), negative ones
    sampling to the left."""
    N, C, H, W = reference_fm.shape
    n = (max_disp + dilation - 1) // dilation
    out = reference_fm.new_zeros(N, 2 * C, n, H, W, dtype=torch.float32)
    for idx, i in enumerate(int(v) for v in torch.linspace(start_disp, start_disp + max_disp - 1, n)):
        if abs(i) >= W:
            continue
        if i > 0:
            out[:, :C, idx, :, i:], out[:, C:, idx, :, i:] = reference_fm[..., i:], target_fm[..., :-i]
        elif i == 0:
            out[:, :C, idx], out[:, C:, idx] = reference_fm, target_fm
        else:
            out[:, :C, idx, :, :i], out[:, C:, idx, :, :i] = reference_fm[..., :i], target_fm[..., -i:]
    return out


def compute_volume(reference, target, maxdisp, side="left"):
    """cost_volume.py:44-56 (pinned by dormant_volumes.npz: the reference run with its device='cuda' zeros redirected to the CPU)."""
    b, c, h, w = reference.shape
    cost = reference.new_zeros(b, c, maxdisp, h, w)
    cost[:, :, 0] = reference - target
    for i in range(1, min(maxdisp, w)):
        if side == "left":
            cost[:, :, i, :, i:] = reference[..., i:] - target[..., :-i]
        else:
            cost[:, :, i, :, :-i] = target[..., i:] - reference[..., :-i]
    return cost


def build_sub_volume(feat_l, feat_r, maxdisp):
    """cost_volume.py:108-117 (pinned by dormant_volumes.npz, see compute_volume)."""
    b, c, h, w = feat_l.shape
    cost = feat_l.new_zeros(b, maxdisp, h, w)
    for i in range(maxdisp):
        cost[:, i, :, :i] = feat_l[..., :i].abs().sum(1)
        if i < w:
            cost[:, i, :, i:] = (feat_l[..., i:] - (feat_r[..., :w - i] if i else feat_r)).abs().sum(1)
    return cost


# ============================================================================= refinement (a13)
def context_upsample(disp_low, up_weights, scale_factor=4):
    """disp_refinement/disp_refinement.py:194-204 (== stereobase/igev_blocks.py:51-63, igev/submodule.py:253-265)."""
    b, c, h, w = disp_low.shape
    unf = F.unfold(disp_low, kernel_size=3, dilation=1, padding=1).reshape(b, -1, h, w)
    unf = F.interpolate(unf, (h * scale_factor, w * scale_factor), mode="nearest")
    return (unf * up_weights).sum(1)


# ============================================================================= training branch (autograd oracle)
def gwc_train_preds(volume, sd, maxdisp, h, w, p="DispProcessor"):
    """gwcnet_disp_processor.py:93-126 with eval-mode (frozen) BatchNorm: the four supervised disparities.
    Built from differentiable torch ops, so torch.autograd on it is the gradient oracle."""
    x = F.relu(convbn3d(volume, sd, p + ".dres0.0"))
    x = F.relu(convbn3d(x, sd, p + ".dres0.2"))
    cost0 = convbn3d(F.relu(convbn3d(x, sd, p + ".dres1.0")), sd, p + ".dres1.2") + x
    out1 = gwc_hourglass(cost0, sd, p + ".dres2")
    out2 = gwc_hourglass(out1, sd, p + ".dres3")
    out3 = gwc_hourglass(out2, sd, p + ".dres4")
    preds = []
    for i, feat in enumerate((cost0, out1, out2, out3)):
        z = F.relu(convbn3d(feat, sd, f"{p}.classif{i}.0"))
        preds.append(upsample_regression(F.conv3d(z, sd[f"{p}.classif{i}.2.weight"], None, 1, 1), maxdisp, h, w, False))
    return preds


def gwc_loss(preds, disp_gt, maxdisp):
    """models/gwcnet/gwcnet.py:42-53 (smooth-L1, weights 0.5/0.5/0.7/1.0, mask 0 < gt < maxdisp)."""
    mask = (disp_gt < maxdisp) & (disp_gt > 0)
    return sum(wt * F.smooth_l1_loss(p_[mask], disp_gt[mask], reduction="mean") for p_, wt in zip(preds, [0.5, 0.5, 0.7, 1.0]))


# ============================================================================= pre-processing (8f #3)
def preprocess_image(img_hwc, pad_size, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """stereo_trans.py:243-267 (RightTopPad, np.pad 'edge' on top/right), :22-29 (HWC->CHW), :32-44 (float32),
    :48-56 (x/255 then (x-mean)/std, torchvision normalize semantics).  img_hwc: numpy [H,W,3]."""
    import numpy as np
    h, w = img_hwc.shape[:2]
    th, tw = pad_size
    h, w = min(h, th), min(w, tw)
    img = np.pad(img_hwc, np.array([[th - h, 0], [0, tw - w], [0, 0]]), "edge")
    t = torch.from_numpy(img.transpose((2, 0, 1)).copy()).to(torch.float32) / 255.0
    m = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    return (t - m) / s


# ============================================================================= geometry-encoding volume (a5)
class GeoEncodingVolume:
    """models/stereobase/gru_blocks.py:170-229 == models/igev/geometry.py:7-66: all-pairs correlation, per-pixel
    rows, avg-pooled pyramid, and per-iteration lookups through F.grid_sample (align_corners=True, zero padding)."""

    def __init__(self, fmap1, fmap2, geo_volume, num_levels=2, radius=4):
        self.num_levels, self.radius = num_levels, radius
        corr = torch.einsum("aijk,aijh->ajkh", fmap1, fmap2)                  # [B,H,W1,W2]
        b, c, d, h, w = geo_volume.shape
        g = geo_volume.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, 1, d)
        cr = corr.reshape(b * h * w, 1, 1, corr.shape[-1])
        self.geo, self.corr = [g], [cr]
        for _ in range(num_levels - 1):
            self.geo.append(F.avg_pool2d(self.geo[-1], [1, 2], stride=[1, 2]))
            self.corr.append(F.avg_pool2d(self.corr[-1], [1, 2], stride=[1, 2]))

    @staticmethod
    def _sample(img, x):
        W = img.shape[-1]
        grid = torch.cat([2 * x / (W - 1) - 1, torch.zeros_like(x)], dim=-1)
        return F.grid_sample(img, grid, align_corners=True)

    def __call__(self, disp, coords):
        r = self.radius
        b, _, h, w = disp.shape
        dx = torch.linspace(-r, r, 2 * r + 1).view(1, 1, 2 * r + 1, 1)
        out = []
        for i in range(self.num_levels):
            x0 = dx + disp.reshape(b * h * w, 1, 1, 1) / 2 ** i
            out.append(self._sample(self.geo[i], x0).view(b, h, w, -1))
            xc = coords.reshape(b * h * w, 1, 1, 1) / 2 ** i - disp.reshape(b * h * w, 1, 1, 1) / 2 ** i + dx
            out.append(self._sample(self.corr[i], xc).view(b, h, w, -1))
        return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


# ----------------------------------------------------------------------------- LightStereo 2-D aggregation (a9)
def _mobile_v2_residual(x, sd, p, stride):
    """MobileV2Residual.forward, stereo/modeling/models/lightstereo/aggregation.py:88-98 (dilation 1)."""
    inp, oup = sd[p + ".pwconv.0.weight"].shape[1], sd[p + ".pwliner.0.weight"].shape[0]
    hid = sd[p + ".dwconv.0.weight"].shape[0]
    f = F.relu6(_bn(F.conv2d(x, sd[p + ".pwconv.0.weight"]), sd, p + ".pwconv.1"))
    f = F.relu6(_bn(F.conv2d(f, sd[p + ".dwconv.0.weight"], None, stride, 1, 1, hid), sd, p + ".dwconv.1"))
    f = _bn(F.conv2d(f, sd[p + ".pwliner.0.weight"]), sd, p + ".pwliner.1")
    return x + f if (stride == 1 and inp == oup) else f


def _ls_attention(cost, x, sd, p):
    """AttentionModule.forward, aggregation.py:118-134."""
    dim = sd[p + ".conv3.weight"].shape[0]
    c = lambda t, n, pad: F.conv2d(t, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1, pad, 1, dim)
    attn = F.conv2d(x, sd[p + ".conv0.weight"], sd[p + ".conv0.bias"])
    a0 = c(c(attn, "conv0_1", (0, 3)), "conv0_2", (3, 0))
    a1 = c(c(attn, "conv1_1", (0, 5)), "conv1_2", (5, 0))
    a2 = c(c(attn, "conv2_1", (0, 10)), "conv2_2", (10, 0))
    attn = attn + a0 + a1 + a2
    attn = F.conv2d(attn, sd[p + ".conv3.weight"], sd[p + ".conv3.bias"])
    return attn * cost


def lightstereo_aggregation(x, features_left, sd, p="", blocks=(1, 2, 4), left_att=True, taps=None):
    """Aggregation.forward, aggregation.py:42-60.  sd: the reference module's state_dict (prefix p)."""
    q = (p + ".") if p else ""
    for i in range(blocks[0]):
        x = _mobile_v2_residual(x, sd, f"{q}conv0.{i}", 1)
    if left_att:
        x = _ls_attention(x, features_left[0], sd, q + "att0")
        if taps is not None:
            taps["att0"] = x
    conv2 = _mobile_v2_residual(x, sd, q + "conv1", 2)
    for i in range(blocks[1] - 1):
        conv2 = _mobile_v2_residual(conv2, sd, f"{q}conv2.{i}", 1)
    if left_att:
        conv2 = _ls_attention(conv2, features_left[1], sd, q + "att2")
    conv4 = _mobile_v2_residual(conv2, sd, q + "conv3", 2)
    for i in range(blocks[2] - 1):
        conv4 = _mobile_v2_residual(conv4, sd, f"{q}conv4.{i}", 1)
    if left_att:
        conv4 = _ls_attention(conv4, features_left[2], sd, q + "att4")
        if taps is not None:
            taps["att4"] = conv4
    up = lambda t, n: _bn(F.conv_transpose2d(t, sd[f"{q}{n}.0.weight"], None, 2, 1, 1), sd, f"{q}{n}.1")
    conv5 = F.relu(up(conv4, "conv5") + _mobile_v2_residual(conv2, sd, q + "redir2", 1))
    conv6 = F.relu(up(conv5, "conv6") + _mobile_v2_residual(x, sd, q + "redir1", 1))
    return conv6


# ----------------------------------------------------------------------------- IGEV / StereoBase update block (8f #4)
def _conv_b(x, sd, p, pad):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], 1, pad)


def _conv_gru(h, cz, cr, cq, xs, sd, p):
    """ConvGRU.forward, stereo/modeling/models/igev/update.py:36-45."""
    x = torch.cat(xs, dim=1)
    hx = torch.cat([h, x], dim=1)
    z = torch.sigmoid(_conv_b(hx, sd, p + ".convz", 1) + cz)
    r = torch.sigmoid(_conv_b(hx, sd, p + ".convr", 1) + cr)
    q = torch.tanh(_conv_b(torch.cat([r * h, x], dim=1), sd, p + ".convq", 1) + cq)
    return (1 - z) * h + z * q


def _motion_encoder(disp, corr, sd, p):
    """BasicMotionEncoder.forward, update.py:83-92."""
    cor = F.relu(_conv_b(corr, sd, p + ".convc1", 0))
    cor = F.relu(_conv_b(cor, sd, p + ".convc2", 1))
    d = F.relu(_conv_b(disp, sd, p + ".convd1", 3))
    d = F.relu(_conv_b(d, sd, p + ".convd2", 1))
    out = F.relu(_conv_b(torch.cat([cor, d], dim=1), sd, p + ".conv", 1))
    return torch.cat([out, disp], dim=1)


def igev_update_block(net, inp, corr, disp, sd, p="", n_gru_layers=3, iter04=True, iter08=True, iter16=True, update=True):
    """BasicMultiUpdateBlock.forward, update.py:129-150 (net is a list of 3 hidden states, finest first)."""
    q = (p + ".") if p else ""
    net = list(net)
    pool2x = lambda t: F.avg_pool2d(t, 3, stride=2, padding=1)
    interp = lambda t, dest: F.interpolate(t, dest.shape[2:], mode="bilinear", align_corners=True)
    if iter16:
        net[2] = _conv_gru(net[2], *inp[2], [pool2x(net[1])], sd, q + "gru16")
    if iter08:
        xs = [pool2x(net[0]), interp(net[2], net[1])] if n_gru_layers > 2 else [pool2x(net[0])]
        net[1] = _conv_gru(net[1], *inp[1], xs, sd, q + "gru08")
    if iter04:
        mf = _motion_encoder(disp, corr, sd, q + "encoder")
        xs = [mf, interp(net[1], net[0])] if n_gru_layers > 1 else [mf]
        net[0] = _conv_gru(net[0], *inp[0], xs, sd, q + "gru04")
    if not update:
        return net
    delta = _conv_b(F.relu(_conv_b(net[0], sd, q + "disp_head.conv1", 1)), sd, q + "disp_head.conv2", 1)
    mask = F.relu(_conv_b(net[0], sd, q + "mask_feat_4.0", 1))
    return net, mask, delta


def lightstereo_cost_stage(features_left, feature_right, sd, max_disp, blocks=(1, 2, 4), left_att=True):
    """lightstereo.py:51-56: correlation volume -> cost_agg -> softmax -> disparity_regression (sd: `cost_agg.*` keys)."""
    D4 = max_disp // 4
    vol = corr_volume(features_left[0], feature_right, D4)
    enc = lightstereo_aggregation(vol, features_left, sd, "cost_agg", blocks=blocks, left_att=left_att)
    prob = F.softmax(enc, dim=1)
    return disparity_regression(prob, D4, keepdim=True), prob, enc


def igev_refine(match_l, match_r, geo_volume, net, inp, init_disp, sd, iters, p="update_block", n_gru_layers=3,
                slow_fast=True, radius=4, num_levels=2):
    """GRU refinement loop, stereo/modeling/models/igev/igev_stereo.py:181-203 (test mode, no final upsample)."""
    geo_fn = GeoEncodingVolume(match_l, match_r, geo_volume, num_levels=num_levels, radius=radius)
    b, _, h, w = match_l.shape
    coords = torch.arange(w).float().reshape(1, 1, w, 1).repeat(b, h, 1, 1)
    disp, net, mask = init_disp, list(net), None
    for _ in range(iters):
        geo_feat = geo_fn(disp, coords)
        if n_gru_layers == 3 and slow_fast:
            net = igev_update_block(net, inp, None, None, sd, p, n_gru_layers, iter16=True, iter08=False, iter04=False, update=False)
        if n_gru_layers >= 2 and slow_fast:
            net = igev_update_block(net, inp, None, None, sd, p, n_gru_layers, iter16=n_gru_layers == 3, iter08=True, iter04=False, update=False)
        net, mask, delta = igev_update_block(net, inp, geo_feat, disp, sd, p, n_gru_layers,
                                             iter16=n_gru_layers == 3, iter08=n_gru_layers >= 2)
        disp = disp + delta
    return disp, mask, net
