"""GPU parity: the HIP engine (through the C ABI) against the golden vectors of the real reference
and against the CPU oracle on seeded inputs.  fp32 everywhere; tolerances are written per test."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import golden, rnd
from openstereo_amd.utils.weights import synth_state_dict, synth_images, synth_tensor

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


def g2(x):
    return (T(x) if isinstance(x, np.ndarray) else x).to(DEV)


def close(a, b, atol=1e-6, rtol=1e-6, what=""):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else b
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: max|err|={err.max():.3e} at {i}: got {a[i]} want {b[i]} "
                             f"({(err > tol).mean() * 100:.2f}% out of tol)")


# ----------------------------------------------------------------------------- volumes
@pytest.mark.parametrize("tag", ["a", "narrow", "k12"])
def test_volume_functions_vs_reference_golden(tag):
    from openstereo_amd import ops
    g = golden("volumes.npz")
    B, C, H, W, D, G = (int(v) for v in g[f"{tag}_meta"])
    L, R = g2(g[f"{tag}_L"]), g2(g[f"{tag}_R"])
    close(ops.build_gwc_volume(L, R, D, G), g[f"{tag}_gwc"], what="build_gwc_volume")
    close(ops.build_concat_volume(L, R, D), g[f"{tag}_concat"], 0, 0, "build_concat_volume")
    close(ops.build_concat_volume(L, R, D, mask_left=False), g[f"{tag}_igev_concat"], 0, 0, "igev concat")
    close(ops.cat_fms(L, R, max_disp=D), g[f"{tag}_psm_cat"], 0, 0, "cat_fms")
    close(ops.correlation_volume(L, R, D), g[f"{tag}_corr"], what="correlation_volume")
    close(ops.build_corr_volume(L, R, D), g[f"{tag}_corr2"], what="build_corr_volume")
    v = ops.build_cost_volume_cl(L, R, G, L[:, :6].contiguous(), R[:, :6].contiguous(), maxdisp=D)
    assert ops.is_cl(v)
    ref = g[f"{tag}_gwcnet_volume"]
    close(v[:, :ref.shape[1]], ref, what="fused NDHWC volume")
    close(ops.to_ncdhw(v, ref.shape[1]), ref, what="fused NDHWC volume via layout kernel")


def test_volume_gwcnet_shape_vs_oracle():
    """GwcNet channel configuration (320ch/40 groups + 12 concat) on a short row block."""
    from openstereo_amd import ops
    from oracle import torch_ref as O
    rng = np.random.default_rng(5)
    lg, rg = (T(rng.normal(0, 1, (1, 320, 3, 50)).astype(np.float32)) for _ in range(2))
    lc, rc = (T(rng.normal(0, 1, (1, 12, 3, 50)).astype(np.float32)) for _ in range(2))
    ref = torch.cat((O.gwc_volume(lg, rg, 48, 40), O.concat_volume(lc, rc, 48)), 1)
    v = ops.build_cost_volume_cl(g2(lg), g2(rg), 40, g2(lc), g2(rc), maxdisp=48)
    assert v.shape == ref.shape and ops.is_cl(v)
    close(v, ref, atol=2e-6, rtol=1e-5, what="gwcnet fused volume")


def test_volume_errors():
    from openstereo_amd import ops, _lib
    x = torch.zeros(1, 10, 4, 8, device=DEV)
    with pytest.raises(AssertionError):
        ops.build_gwc_volume(x, x, 4, 3)                  # C % groups != 0 (cost_volume.py:61)
    with pytest.raises(_lib.EngineError):
        ops.build_gwc_volume(x.cpu(), x.cpu(), 4, 2)      # no CPU path
    with pytest.raises(AssertionError):
        ops.disparity_regression(torch.zeros(1, 4, 4, device=DEV), 4)


def test_layout_roundtrip():
    from openstereo_amd import ops
    x = torch.randn(2, 7, 3, 5, 9, device=DEV)
    y = ops.to_cl(x)
    assert y.shape[1] == 8 and ops.is_cl(y)
    assert torch.equal(y[:, :7], x) and torch.count_nonzero(y[:, 7:]) == 0
    assert torch.equal(ops.to_ncdhw(y, 7), x)


# ----------------------------------------------------------------------------- regression
def test_regression_vs_reference_golden():
    from openstereo_amd import ops
    g = golden("regression.npz")
    prob, cost = g2(g["prob"]), g2(g["cost"])
    close(ops.disparity_regression(prob, 12), g["reg_keep"], atol=1e-5, what="disparity_regression keepdim")
    close(ops.disparity_regression(prob, 12, keepdim=False), g["reg_nokeep"], atol=1e-5, what="gwc regression")
    close(ops.FasterSoftArgmin(12).to(DEV)(cost), g["faster_softargmin"], atol=1e-5, what="FasterSoftArgmin")
    d, p = ops.softmax_disparity_regression(cost, keepdim=False, return_prob=True)
    close(p, g["prob"], atol=1e-6, what="softmax prob")
    close(ops.upsample_softargmin(g2(g["low"]), 24, 20, 28, False), g["up_false"], atol=2e-5, what="upsample F")
    close(ops.upsample_softargmin(g2(g["low"]), 24, 20, 28, True), g["up_true"], atol=2e-5, what="upsample T")
    close(ops.upsample_softargmin(g2(g["low2"]), 17, 13, 21, False), g["up_odd"], atol=2e-5, what="upsample odd")
    with pytest.raises(ValueError):
        ops.FasterSoftArgmin(12)(torch.zeros(1, 1, 12, 4, 4, device=DEV))


# ----------------------------------------------------------------------------- single conv layers
def _bn_for(c, seed, name):
    bn = nn.BatchNorm3d(c)
    bn.load_state_dict({k: synth_tensor(f"{name}.{k}", v.shape, seed) for k, v in bn.state_dict().items()})
    return bn.eval()


CONV_CASES = [
    # name, Ci, Co, k, stride, pad, dil, (D,H,W), act, residual
    ("32-32 s1", 32, 32, 3, 1, 1, 1, (6, 9, 20), "relu", False),
    ("64-32 s1 ragged", 64, 32, 3, 1, 1, 1, (5, 7, 11), "relu", True),
    ("32-64 s2", 32, 64, 3, 2, 1, 1, (8, 12, 20), "relu", False),
    ("64-128 s2 odd", 64, 128, 3, 2, 1, 1, (6, 10, 14), "relu", False),
    ("128-128 s1", 128, 128, 3, 1, 1, 1, (3, 6, 10), "relu", False),
    ("64-64 s1", 64, 64, 3, 1, 1, 1, (4, 8, 8), "none", True),
    ("1x1 32-32", 32, 32, 1, 1, 0, 1, (4, 5, 9), "none", False),
    ("1x1 64-64", 64, 64, 1, 1, 0, 1, (3, 4, 7), "none", False),
    ("24-48 leaky", 24, 48, 3, 1, 1, 1, (4, 6, 10), "leaky", False),
    ("8-16", 8, 16, 3, 1, 1, 1, (5, 6, 7), "relu", False),
    ("48-24 s2", 48, 24, 3, 2, 1, 1, (8, 8, 12), "leaky", False),
]


PRECS = ["f32", "f16x3"]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv3d_bn_act_vs_oracle(case, prec):
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d
    name, Ci, Co, k, s, p, dil, (D, H, W), act, use_res = case
    conv = nn.Conv3d(Ci, Co, k, s, p, dil, bias=False)
    conv.weight.data = synth_tensor(name + ".w", conv.weight.shape, 1)
    bn = _bn_for(Co, 2, name)
    x = T(np.random.default_rng(3).normal(0, 1, (2, Ci, D, H, W)).astype(np.float32))
    with torch.no_grad():
        ref = bn(conv(x))
        res = torch.randn(ref.shape, generator=torch.Generator().manual_seed(4)) if use_res else None
        if res is not None:
            ref = ref + res
        ref = {"relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.01), "none": lambda t: t}[act](ref)
    pc = PackedConv3d(conv.to(DEV), bn.to(DEV), {"none": 0, "relu": 1, "leaky": 2}[act], 0.01, precision=prec)
    y = pc(ops.to_cl(x.to(DEV)), residual=None if res is None else ops.to_cl(res.to(DEV)))
    assert ops.is_cl(y)
    close(y[:, :Co], ref, atol=2e-5, rtol=2e-5, what=f"{name} [{prec}]")


DECONV_CASES = [("k3 128-64", 128, 64, 3, 1, 1, (3, 5, 7)), ("k3 64-32", 64, 32, 3, 1, 1, (4, 6, 9)),
                ("k4 48-24", 48, 24, 4, 1, 0, (3, 4, 6)), ("k4 16-8", 16, 8, 4, 1, 0, (4, 5, 5))]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("case", DECONV_CASES, ids=[c[0] for c in DECONV_CASES])
def test_deconv3d_bn_residual_vs_oracle(case, prec):
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d
    name, Ci, Co, k, p, op, (D, H, W) = case
    dc = nn.ConvTranspose3d(Ci, Co, k, stride=2, padding=p, output_padding=op, bias=False)
    dc.weight.data = synth_tensor(name + ".w", dc.weight.shape, 1)
    bn = _bn_for(Co, 2, name)
    x = T(np.random.default_rng(3).normal(0, 1, (2, Ci, D, H, W)).astype(np.float32))
    with torch.no_grad():
        up = bn(dc(x))
        res = torch.randn(up.shape, generator=torch.Generator().manual_seed(4))
        ref = F.relu(up + res)
    pc = PackedConv3d(dc.to(DEV), bn.to(DEV), 1, precision=prec)
    y = pc(ops.to_cl(x.to(DEV)), residual=ops.to_cl(res.to(DEV)))
    close(y[:, :Co], ref, atol=2e-5, rtol=2e-5, what=f"{name} [{prec}]")


def test_conv2d_as_flat_conv3d_dilated():
    """2-D layers (D=1, 1x3x3 kernel, dilation 2) run through the same kernel."""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d
    conv = nn.Conv3d(32, 32, (1, 3, 3), 1, (0, 2, 2), (1, 2, 2), bias=False)
    conv.weight.data = synth_tensor("flat.w", conv.weight.shape, 1)
    x = T(np.random.default_rng(3).normal(0, 1, (2, 32, 1, 21, 37)).astype(np.float32))
    with torch.no_grad():
        ref = conv(x)
    y = PackedConv3d(conv.to(DEV))(ops.to_cl(x.to(DEV)))
    close(y, ref, atol=2e-5, rtol=2e-5, what="flat dilated conv")


def test_small_co_classifier_vs_oracle():
    from openstereo_amd import ops
    from openstereo_amd.engine import SmallCoConv3d
    conv = nn.Conv3d(32, 1, 3, 1, 1, bias=False)
    conv.weight.data = synth_tensor("cls.w", conv.weight.shape, 1)
    x = T(np.random.default_rng(3).normal(0, 1, (2, 32, 5, 7, 9)).astype(np.float32))
    with torch.no_grad():
        ref = conv(x)
    y = SmallCoConv3d(conv.to(DEV))(ops.to_cl(x.to(DEV)))
    close(y, ref, atol=2e-4, rtol=2e-5, what="classifier head")


@pytest.mark.parametrize("shape", [(2, 19, 37, 45), (1, 3, 16, 16), (1, 1, 5, 70), (3, 24, 33, 17)])
def test_classifier_marching_form_vs_oracle(shape):
    """The 32 -> 1 classifier shape takes the d-marching kernel (conv3d.hip, classifier_march_kernel): ragged 16x16 tiles, several D
    segments (19 planes -> 2 segments of 10), D smaller than the halo, bias and residual."""
    from openstereo_amd import ops
    from openstereo_amd.engine import SmallCoConv3d
    B, D, H, W = shape
    conv = nn.Conv3d(32, 1, 3, 1, 1, bias=True)
    conv.weight.data = synth_tensor("clsm.w", conv.weight.shape, 1)
    conv.bias.data = synth_tensor("clsm.b", conv.bias.shape, 1)
    rng = np.random.default_rng(11)
    x = T(rng.normal(0, 1, (B, 32, D, H, W)).astype(np.float32))
    r = T(rng.normal(0, 1, (B, 1, D, H, W)).astype(np.float32))
    with torch.no_grad():
        ref = conv(x) + r
    y = SmallCoConv3d(conv.to(DEV))(ops.to_cl(x.to(DEV)), residual=ops.to_cl(r.to(DEV), pad_to=1))
    close(y, ref, atol=2e-4, rtol=2e-5, what=f"classifier head (marching form) {shape}")


# ----------------------------------------------------------------------------- GwcNet stages / model
def test_gwc_hourglass_vs_reference_golden():
    from openstereo_amd.models.gwcnet import Hourglass
    g = golden("gwc_hourglass.npz")
    hg = Hourglass(8)
    hg.load_state_dict(synth_state_dict(hg, seed=3))
    hg = hg.to(DEV).eval()
    close(hg(g2(g["x"])), g["y"], atol=2e-5, rtol=2e-5, what="Hourglass(8) drop-in forward")


def test_gwc_disp_processor_vs_reference_golden():
    from openstereo_amd import ops
    from openstereo_amd.models.gwcnet import GwcDispProcessor
    g = golden("gwc_disp.npz")
    dp = GwcDispProcessor(maxdisp=32)
    dp.load_state_dict(synth_state_dict(dp, seed=4))
    dp = dp.to(DEV).eval()
    vol = g2(g["volume"])
    p = dp._pack()
    x = ops.to_cl(vol)
    cost0 = p["d02"](p["d00"](x))
    cost0 = p["d12"](p["d10"](cost0), residual=cost0)
    close(cost0, g["cost0"], atol=3e-5, rtol=3e-5, what="cost0")
    out1 = dp.dres2.forward_cl(cost0)
    close(out1, g["out1"], atol=5e-5, rtol=5e-5, what="out1")
    cost3 = dp.aggregate_cl(x)
    close(cost3, g["cost3"], atol=5e-4, rtol=1e-4, what="cost3")
    disp = dp({"cost_volume": vol, "left": torch.zeros(1, 3, 32, 64, device=DEV)})["inference_disp"]["disp_est"]
    epe = (disp.cpu().numpy() - g["disp"]).__abs__().mean()
    assert epe < 1e-3, f"EPE {epe}"
    close(disp, g["disp"], atol=5e-3, what="disp")


def _gwcnet():
    from openstereo_amd.models.gwcnet import GwcNet
    net = GwcNet()
    net.load_state_dict(synth_state_dict(net, seed=0))
    return net.to(DEV).eval()


def test_gwcnet_small_vs_reference_golden():
    g = golden("gwcnet_small.npz")
    net = _gwcnet()
    L, R = synth_images(1, 64, 128, seed=1)
    with torch.no_grad():
        feats = net.Backbone({"left": L.to(DEV), "right": R.to(DEV)})          # engine backbone, NCHW contract
        disp = net({"left": L.to(DEV), "right": R.to(DEV)})["disp_pred"]       # fused NHWC path
    close(feats["ref_feature"]["gwc_feature"], g["left_gwc"], atol=1e-4, rtol=1e-4, what="engine backbone gwc feature (left)")
    close(feats["tgt_feature"]["gwc_feature"], g["right_gwc"], atol=1e-4, rtol=1e-4, what="engine backbone gwc feature (right)")
    close(feats["ref_feature"]["concat_feature"], g["left_cat"], atol=1e-4, rtol=1e-4, what="engine backbone concat feature")
    epe = (disp.cpu().numpy() - g["disp"]).__abs__().mean()
    assert disp.shape == (1, 64, 128)
    assert epe < 1e-3, f"EPE {epe}"
    # the PyTorch-ROCm (MIOpen) backbone stays available and must agree too
    net.Backbone.use_engine = False
    with torch.no_grad():
        disp2 = net({"left": L.to(DEV), "right": R.to(DEV)})["disp_pred"]
    assert np.abs(disp2.cpu().numpy() - g["disp"]).mean() < 1e-3


def test_gwcnet_engine_from_reference_features_small():
    """Engine-only parity: feed the reference's own feature maps, compare cost3 + disparity."""
    from openstereo_amd import ops
    g = golden("gwcnet_small.npz")
    net = _gwcnet()
    vol = ops.build_cost_volume_cl(g2(g["left_gwc"]), g2(g["right_gwc"]), 40, g2(g["left_cat"]), g2(g["right_cat"]), 48)
    cost3 = net.DispProcessor.aggregate_cl(vol)
    close(cost3, g["cost3"], atol=1e-3, rtol=1e-4, what="cost3 from reference features")
    disp = ops.upsample_softargmin(cost3, 192, 64, 128)
    epe = (disp.cpu().numpy() - g["disp"]).__abs__().mean()
    assert epe < 1e-3, f"EPE {epe}"


@pytest.mark.parametrize("prec", PRECS)
def test_gwcnet_full_size_vs_reference_golden(prec):
    """BASELINE configs[1]: 540x960 padded to 544x960, D=192; disparity within 1e-3 EPE of the
    reference CPU path (golden produced by the real reference) -- in both arithmetic modes."""
    from openstereo_amd import engine
    g = golden("gwcnet_full_disp.npz")
    old = engine.get_precision()
    engine.set_precision(prec)
    try:
        _full_size_check(g, prec)
    finally:
        engine.set_precision(old)


def _full_size_check(g, prec):
    net = _gwcnet()
    L, R = synth_images(1, 544, 960, seed=1)
    with torch.no_grad():
        disp = net({"left": L.to(DEV), "right": R.to(DEV)})["disp_pred"]
    torch.cuda.synchronize()
    d = disp.cpu().numpy()
    assert np.isfinite(d).all()
    err = np.abs(d - g["disp"])
    print(f"[{prec}] full-size EPE {err.mean():.3e}, max {err.max():.3e}, >0.01px: {(err > 1e-2).mean() * 100:.4f}%")
    assert err.mean() < 1e-3, f"EPE {err.mean()}"


# ----------------------------------------------------------------------------- PSMNet (BASELINE configs[0])
def test_psmnet_256x512_vs_reference_golden():
    from openstereo_amd.models.psmnet import PSMNet, _Cfg
    g = golden("psmnet_256x512.npz")
    net = PSMNet(_Cfg(MAX_DISP=64))
    net.load_state_dict(synth_state_dict(net, seed=0, head_gain=3.0), strict=False)
    net = net.to(DEV).eval()
    L, R = synth_images(1, 256, 512, seed=1, max_shift=16.0)
    with torch.no_grad():
        inputs = {"left": L.to(DEV), "right": R.to(DEV)}
        out = net(inputs)
    close(inputs["ref_feature"], g["left_feature"], atol=2e-4, rtol=2e-4, what="PSM backbone feature")
    assert out["disp_pred"].shape == (1, 256, 512) and len(out["train_preds"]) == 3
    for d, k in zip(out["train_preds"], ("disp1", "disp2", "disp3")):
        epe = np.abs(d.cpu().numpy() - g[k]).mean()
        assert epe < 1e-3, (k, epe)


def test_psmnet_engine_from_reference_features():
    """Engine-only: reference feature maps in, aggregator tap + three disparities out."""
    from openstereo_amd import ops
    from openstereo_amd.models.psmnet import PSMNet, _Cfg
    g = golden("psmnet_256x512.npz")
    net = PSMNet(_Cfg(MAX_DISP=64))
    net.load_state_dict(synth_state_dict(net, seed=0, head_gain=3.0), strict=False)
    net = net.to(DEV).eval()
    inputs = {"left": torch.zeros(1, 3, 256, 512, device=DEV), "ref_feature": g2(g["left_feature"]),
              "tgt_feature": g2(g["right_feature"])}
    inputs.update(net.CostProcessor(inputs))
    for d, k in zip(net.DispProcessor(inputs), ("disp1", "disp2", "disp3")):
        epe = np.abs(d.cpu().numpy() - g[k]).mean()
        assert epe < 2e-4, (k, epe)


def test_psm_hourglass_dropin_vs_oracle():
    from openstereo_amd.models.psmnet import Hourglass
    from oracle import torch_ref as O
    hg = Hourglass(8)
    sd = synth_state_dict(hg, seed=5)
    hg.load_state_dict(sd)
    x = T(np.random.default_rng(1).normal(0, 1, (1, 8, 8, 8, 12)).astype(np.float32))
    pre_in = T(np.random.default_rng(2).normal(0, 1, (1, 16, 4, 4, 6)).astype(np.float32))
    post_in = T(np.random.default_rng(3).normal(0, 1, (1, 16, 4, 4, 6)).astype(np.float32))
    sdp = {"h." + k: v for k, v in sd.items()}
    hg = hg.to(DEV).eval()
    for (a, b) in ((None, None), (pre_in, post_in)):
        ref = O.psm_hourglass(x, sdp, "h", a, b)
        got = hg(x.to(DEV), None if a is None else a.to(DEV), None if b is None else b.to(DEV))
        for r, o, nm in zip(ref, got, ("out", "pre", "post")):
            close(o, r, atol=3e-5, rtol=3e-5, what=f"psm hourglass {nm}")


# ----------------------------------------------------------------------------- StereoBase / IGEV aggregation (a8)
def test_stereobase_hourglass_vs_reference_golden():
    from openstereo_amd.models.igev_style import Hourglass
    g = golden("stereobase_hourglass.npz")
    hg = Hourglass(24, [96, 64, 192, 120])
    hg.load_state_dict(synth_state_dict(hg, seed=6))
    hg = hg.to(DEV).eval()
    f = [None, g2(g["f1"]), g2(g["f2"]), g2(g["f3"])]
    with torch.no_grad():
        y, y1, y2 = hg(g2(g["x"]), f, return_multi=True)
        y_single = hg(g2(g["x"]), f)
    close(y2, g["y2"], atol=3e-5, rtol=3e-5, what="stereobase hourglass conv2 (1/16)")
    close(y1, g["y1"], atol=3e-5, rtol=3e-5, what="stereobase hourglass conv1 (1/8)")
    close(y, g["y"], atol=3e-5, rtol=3e-5, what="stereobase hourglass out")
    close(y_single, g["y"], atol=3e-5, rtol=3e-5, what="stereobase hourglass out (single)")


def test_igev_hourglass_vs_reference_golden():
    from openstereo_amd.models.igev_style import hourglass
    g = golden("igev_hourglass.npz")
    hg = hourglass(8)
    hg.load_state_dict(synth_state_dict(hg, seed=7))
    hg = hg.to(DEV).eval()
    f = [None, g2(g["f1"]), g2(g["f2"]), g2(g["f3"])]
    with torch.no_grad():
        y = hg(g2(g["x"]), f)
    close(y, g["y"], atol=3e-5, rtol=3e-5, what="igev hourglass")


def test_stereobase_cost_stage_vs_oracle():
    """gwc(8 groups of 12) + concat(2x8) volume -> Hourglass(24) -> classifier -> softmax -> regression."""
    from openstereo_amd.models.igev_style import StereoBaseCostStage
    from oracle import torch_ref as O
    st = StereoBaseCostStage(max_disp=64, num_groups=8, concat_channels=8, backbone_channels=[96, 64, 192, 120])
    sd = synth_state_dict(st, seed=8, head_gain=20.0)
    st.load_state_dict(sd)
    r = np.random.default_rng(9)
    mk = lambda *s: T(r.normal(0, 1, s).astype(np.float32))
    ml, mr, cl, cr = mk(1, 96, 16, 40), mk(1, 96, 16, 40), mk(1, 8, 16, 40), mk(1, 8, 16, 40)
    feats = [None, mk(1, 64, 8, 20), mk(1, 192, 4, 10), mk(1, 120, 2, 5)]
    with torch.no_grad():
        ref_disp, ref_prob, ref_geo = O.stereobase_cost_stage(ml, mr, cl, cr, feats, sd, 64, 8)
        st = st.to(DEV).eval()
        out = st(ml.to(DEV), mr.to(DEV), cl.to(DEV), cr.to(DEV), [None] + [f.to(DEV) for f in feats[1:]])
    close(out["geo_encoding_volume"], ref_geo, atol=5e-5, rtol=5e-5, what="geo encoding volume")
    close(out["prob"], ref_prob, atol=2e-5, rtol=1e-4, what="prob")
    close(out["init_disp"], ref_disp, atol=5e-4, what="init disp")
    assert float(ref_disp.std()) > 0.3        # not a degenerate (uniform-softmax) case


def test_volume_from_channels_last_features():
    """NHWC feature maps (engine backbone layout) -> same volume as the NCHW path."""
    from openstereo_amd import ops
    r = np.random.default_rng(11)
    B, H, W = 2, 5, 37
    gw = T(r.normal(0, 1, (2 * B, 320, H, W)).astype(np.float32)).to(DEV)
    ct = T(r.normal(0, 1, (2 * B, 12, H, W)).astype(np.float32)).to(DEV)
    ref = ops.build_cost_volume_cl(gw[:B], gw[B:], 40, ct[:B], ct[B:], maxdisp=24)
    got = ops.build_cost_volume_from_cl(ops.to_cl(gw.unsqueeze(2)), 40, ops.to_cl(ct.unsqueeze(2)), B, 24)
    assert torch.equal(ref, got)


def test_context_upsample_vs_reference_golden():
    from openstereo_amd import ops
    g = golden("context_upsample.npz")
    dl, wt, lg = g2(g["disp_low"]), g2(g["weights"]), g2(g["logits"])
    close(ops.context_upsample(dl * 4., wt), g["out"], atol=2e-5, what="context_upsample")
    close(ops.context_upsample(dl, lg, softmax_weights=True, gain=4.0), g["out"], atol=2e-5, what="fused softmax+gain")
    close(ops.context_upsample(dl[:, :, :3, :4].contiguous(), wt[:, :, :6, :8].contiguous(), scale_factor=2),
          g["out_s2"], atol=2e-5, what="scale 2")


def test_preprocess_pair_vs_reference_golden():
    """RightTopPad + transpose + /255 + normalise fused on device vs the output of the reference's own transform classes
    (tests/golden/preprocess.npz, stereo_trans.py:22-56,243-267): uint8 and float32 HWC inputs, NCHW and NHWC4 outputs."""
    from openstereo_amd import ops
    g = golden("preprocess.npz")
    for dt in (np.uint8, np.float32):
        L, R = g["left_u8"].astype(dt), g["right_u8"].astype(dt)
        l, rr = ops.preprocess_pair(T(L).to(DEV), T(R).to(DEV), (32, 48))
        close(l[0], g["left"], atol=1e-6, rtol=1e-6, what=f"left {dt.__name__} vs reference transforms")
        close(rr[0], g["right"], atol=1e-6, rtol=1e-6, what=f"right {dt.__name__} vs reference transforms")
        cl = ops.preprocess_pair(T(L).to(DEV), T(R).to(DEV), (32, 48), channels_last=True)
        close(cl[1, :3, 0], g["right"], atol=1e-6, rtol=1e-6, what="NHWC4 right")


def test_preprocess_pair_vs_oracle():
    """RightTopPad + transpose + normalise fused on device vs the numpy/torch restatement of the transform chain."""
    from openstereo_amd import ops
    from oracle import torch_ref as O
    r = np.random.default_rng(12)
    for dt in (np.uint8, np.float32):
        L = r.integers(0, 256, (27, 45, 3)).astype(dt)
        R = r.integers(0, 256, (27, 45, 3)).astype(dt)
        l, rr = ops.preprocess_pair(T(L).to(DEV), T(R).to(DEV), (32, 48))
        close(l[0], O.preprocess_image(L, (32, 48)), atol=1e-6, rtol=1e-6, what=f"left {dt.__name__}")
        close(rr[0], O.preprocess_image(R, (32, 48)), atol=1e-6, rtol=1e-6, what=f"right {dt.__name__}")
        cl = ops.preprocess_pair(T(L).to(DEV), T(R).to(DEV), (32, 48), channels_last=True)
        assert ops.is_cl(cl) and cl.shape == (2, 4, 1, 32, 48)
        close(cl[0, :3, 0], O.preprocess_image(L, (32, 48)), atol=1e-6, rtol=1e-6, what="NHWC4 left")
        assert float(cl[:, 3].abs().max()) == 0.0


def test_geo_encoding_volume_vs_reference_golden():
    """All-pairs correlation, pyramid and the fused per-iteration lookup (a5) vs the reference class."""
    from openstereo_amd.geometry import CombinedGeoEncodingVolume
    g = golden("geo_encoding.npz")
    close(CombinedGeoEncodingVolume.corr(g2(g["f1"]), g2(g["f2"])), g["corr"], atol=2e-6, rtol=1e-5, what="all-pairs corr")
    gev = CombinedGeoEncodingVolume(g2(g["f1"]), g2(g["f2"]), g2(g["geo"]), num_levels=2, radius=4)
    out = gev(g2(g["disp"]), g2(g["coords"]))
    assert out.shape == g["lookup"].shape
    close(out, g["lookup"], atol=2e-5, rtol=1e-5, what="lookup")
    close(gev(g2(g["disp"]) * 2.5 + 1.0, g2(g["coords"])), g["lookup2"], atol=2e-5, rtol=1e-5, what="lookup (large disp, out-of-range taps)")


# ----------------------------------------------------------------------------- LightStereo 2-D aggregation (a9)
DW_CASES = [
    # name, C, (kh, kw), stride, (ph, pw), (H, W), bias, bn, act, add
    ("dw3x3 s1 bn relu6", 192, (3, 3), 1, (1, 1), (13, 22), False, True, "relu6", False),
    ("dw3x3 s2 bn relu6", 96, (3, 3), 2, (1, 1), (14, 21), False, True, "relu6", False),
    ("strip 1x7 bias", 48, (1, 7), 1, (0, 3), (9, 30), True, False, "none", False),
    ("strip 21x1 bias add", 48, (21, 1), 1, (10, 0), (25, 11), True, False, "none", True),
    ("strip 1x11 bias add", 96, (1, 11), 1, (0, 5), (6, 17), True, False, "none", True),
    ("strip 7x1 bias (column runs)", 48, (7, 1), 1, (3, 0), (10, 13), True, False, "none", False),
    ("strip 11x1 bias add (column runs, ragged rows)", 192, (11, 1), 1, (5, 0), (6, 9), True, False, "none", True),
    ("strip 21x1 bias (column runs, short map)", 96, (21, 1), 1, (10, 0), (3, 7), True, False, "none", False),
]


@pytest.mark.parametrize("case", DW_CASES, ids=[c[0] for c in DW_CASES])
def test_depthwise_conv2d_vs_torch(case):
    from openstereo_amd.engine import DepthwiseConv2d
    from openstereo_amd.models.lightstereo import nchw_to_cl, cl_to_nchw
    name, C, k, s, p, (H, W), bias, use_bn, act, use_add = case
    conv = nn.Conv2d(C, C, k, s, p, groups=C, bias=bias)
    conv.weight.data = synth_tensor(name + ".w", conv.weight.shape, 1)
    if bias:
        conv.bias.data = synth_tensor(name + ".bias", conv.bias.shape, 1)
    bn = None
    if use_bn:
        bn = nn.BatchNorm2d(C)
        bn.load_state_dict({k_: synth_tensor(f"{name}.{k_}", v.shape, 2) for k_, v in bn.state_dict().items()})
        bn.eval()
    x = T(np.random.default_rng(3).normal(0, 2, (2, C, H, W)).astype(np.float32))
    with torch.no_grad():
        ref = conv(x)
        if bn is not None:
            ref = bn(ref)
        ref = {"relu6": F.relu6, "none": lambda t: t}[act](ref)
        add = torch.randn(ref.shape, generator=torch.Generator().manual_seed(4)) if use_add else None
        if add is not None:
            ref = ref + add
    layer = DepthwiseConv2d(conv.to(DEV), None if bn is None else bn.to(DEV), {"none": 0, "relu6": 3}[act])
    y = layer(nchw_to_cl(x.to(DEV)), add=None if add is None else nchw_to_cl(add.to(DEV)))
    close(cl_to_nchw(y), ref, atol=1e-5, rtol=1e-5, what=name)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("case", [("192-96", 192, 96, (5, 9)), ("96-48", 96, 48, (8, 16)), ("24-12 ragged", 24, 12, (7, 19))],
                         ids=lambda c: c[0])
def test_deconv2d_bn_residual_vs_torch(case, prec):
    """nn.ConvTranspose2d(k=3, s=2, p=1, op=1) + BN + residual + ReLU: the D = 1 parity-class kernel."""
    from openstereo_amd.engine import PackedConv3d
    from openstereo_amd.models.lightstereo import nchw_to_cl, cl_to_nchw
    name, Ci, Co, (H, W) = case
    dc = nn.ConvTranspose2d(Ci, Co, 3, padding=1, output_padding=1, stride=2, bias=False)
    dc.weight.data = synth_tensor(name + ".w", dc.weight.shape, 1)
    bn = nn.BatchNorm2d(Co)
    bn.load_state_dict({k: synth_tensor(f"{name}.{k}", v.shape, 2) for k, v in bn.state_dict().items()})
    bn.eval()
    x = T(np.random.default_rng(3).normal(0, 1, (2, Ci, H, W)).astype(np.float32))
    with torch.no_grad():
        up = bn(dc(x))
        res = torch.randn(up.shape, generator=torch.Generator().manual_seed(4))
        ref = F.relu(up + res)
    pc = PackedConv3d(dc.to(DEV), bn.to(DEV), 1, precision=prec)
    y = pc(nchw_to_cl(x.to(DEV)), residual=nchw_to_cl(res.to(DEV)))
    close(cl_to_nchw(y, Co), ref, atol=2e-5, rtol=2e-5, what=f"deconv2d {name} [{prec}]")


@pytest.mark.parametrize("prec", PRECS)
def test_pointwise_relu6_and_raw_gate(prec):
    """1x1 Conv2d + BN + ReLU6 (MobileV2Residual.pwconv) and conv3(attn) * cost (AttentionModule)."""
    from openstereo_amd.engine import PackedConv3d, ACT_RELU6
    from openstereo_amd.models.lightstereo import nchw_to_cl, cl_to_nchw
    conv = nn.Conv2d(48, 192, 1, bias=False)
    conv.weight.data = synth_tensor("pw.w", conv.weight.shape, 1) * 4.0     # push part of the outputs beyond 6
    bn = nn.BatchNorm2d(192)
    bn.load_state_dict({k: synth_tensor(f"pw.{k}", v.shape, 2) for k, v in bn.state_dict().items()})
    bn.eval()
    x = T(np.random.default_rng(3).normal(0, 1, (2, 48, 9, 14)).astype(np.float32))
    with torch.no_grad():
        ref = F.relu6(bn(conv(x)))
    assert (ref == 6).any() and (ref == 0).any()
    y = PackedConv3d(conv.to(DEV), bn.to(DEV), ACT_RELU6, precision=prec)(nchw_to_cl(x.to(DEV)))
    close(cl_to_nchw(y), ref, atol=2e-5, rtol=2e-5, what=f"pw relu6 [{prec}]")

    c3 = nn.Conv2d(48, 48, 1)
    c3.weight.data = synth_tensor("c3.w", c3.weight.shape, 1)
    c3.bias.data = synth_tensor("c3.bias", c3.bias.shape, 1)
    cost = T(np.random.default_rng(5).normal(0, 2, (2, 48, 9, 14)).astype(np.float32))
    with torch.no_grad():
        ref = c3(x) * cost
    cc = nchw_to_cl(cost.to(DEV))
    gate = cc.permute(0, 2, 3, 4, 1).reshape(2, 9, 14, 48)
    y = PackedConv3d(c3.to(DEV), precision=prec)(nchw_to_cl(x.to(DEV)), gate=gate, gate_raw=True)
    close(cl_to_nchw(y), ref, atol=4e-5, rtol=2e-5, what=f"raw gate [{prec}]")


@pytest.mark.parametrize("prec", PRECS)
def test_lightstereo_aggregation_vs_reference_golden(prec):
    """a9 end to end: engine Aggregation (drop-in forward, NCHW in/out) vs the real reference's output."""
    from conftest import lightstereo_case
    from openstereo_amd import engine
    g = golden("lightstereo_agg.npz")
    agg, sd, x, feats = lightstereo_case()
    old = engine.get_precision()
    engine.set_precision(prec)
    try:
        agg = agg.to(DEV)
        with torch.no_grad():
            y = agg(x.to(DEV), [f.to(DEV) for f in feats])[0]
            a0 = agg.att0(_ls_conv0(agg, x.to(DEV)), feats[0].to(DEV))
    finally:
        engine.set_precision(old)
    ref = g["y"]
    tol = 2e-4 * max(1.0, float(np.abs(ref).max()))
    close(a0, g["att0"], atol=2e-4 * max(1.0, float(np.abs(g["att0"]).max())), rtol=1e-4, what=f"att0 [{prec}]")
    close(y, ref, atol=tol, rtol=1e-4, what=f"LightStereo aggregation [{prec}]")


def _ls_conv0(agg, x):
    from openstereo_amd.models.lightstereo import nchw_to_cl, cl_to_nchw
    t = nchw_to_cl(x)
    for blk in agg.conv0:
        t = blk.forward_cl(t)
    return cl_to_nchw(t, 48)


def test_lightstereo_aggregation_kitti15_size_vs_oracle():
    """BASELINE configs[3]: LightStereo 2-D aggregation at KITTI15 size (375x1242 padded to 384x1248,
    quarter resolution 96x312, D/4 = 48) -- engine (f16x3) vs the CPU oracle on the same seeded inputs."""
    from conftest import lightstereo_case, rnd
    from oracle import torch_ref as O
    from openstereo_amd import engine
    agg, sd, _, _ = lightstereo_case()
    x = rnd((1, 48, 96, 312), 61).abs()
    feats = [rnd((1, 24, 96, 312), 62), rnd((1, 32, 48, 156), 63), rnd((1, 96, 24, 78), 64)]
    with torch.no_grad():
        ref = O.lightstereo_aggregation(x, feats, sd)
    old = engine.get_precision()
    engine.set_precision("f16x3")
    try:
        agg = agg.to(DEV)
        with torch.no_grad():
            y = agg(x.to(DEV), [f.to(DEV) for f in feats])[0]
    finally:
        engine.set_precision(old)
    close(y, ref, atol=2e-4 * max(1.0, ref.abs().max().item()), rtol=1e-4, what="LightStereo aggregation @96x312")


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("chans", [(64, 32), (128, 64)], ids=["conv6+redir1", "conv5+redir2"])
@pytest.mark.parametrize("shape", [(3, 5, 7), (4, 6, 9)], ids=["3x5x7", "4x6x9"])
def test_deconv3d_fused_redir_matches_two_launches(shape, chans, prec):
    """relu(BN(deconv(c5)) + BN_r(conv1x1x1(x))) with the redir branch computed inside the transposed
    conv's epilogue: bit-identical to deconv + separate 1x1x1 launch, and within tolerance of torch."""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d
    D, H, W = shape
    Ci, Co = chans
    dc = nn.ConvTranspose3d(Ci, Co, 3, stride=2, padding=1, output_padding=1, bias=False)
    dc.weight.data = synth_tensor("fr.dc.w", dc.weight.shape, 1)
    rc = nn.Conv3d(Co, Co, 1, bias=False)
    rc.weight.data = synth_tensor("fr.rc.w", rc.weight.shape, 1)
    bn, bnr = _bn_for(Co, 2, "fr.bn"), _bn_for(Co, 3, "fr.bnr")
    c5 = T(np.random.default_rng(3).normal(0, 1, (2, Ci, D, H, W)).astype(np.float32))
    x = T(np.random.default_rng(4).normal(0, 1, (2, Co, 2 * D, 2 * H, 2 * W)).astype(np.float32))
    with torch.no_grad():
        ref = F.relu(bn(dc(c5)) + bnr(rc(x)))
    pd = PackedConv3d(dc.to(DEV), bn.to(DEV), 1, precision=prec)
    pr = PackedConv3d(rc.to(DEV), bnr.to(DEV), 0, precision=prec)
    c5c, xc = ops.to_cl(c5.to(DEV)), ops.to_cl(x.to(DEV))
    two = pd(c5c, residual=pr(xc))
    fused = pd(c5c, redir=(pr, xc))
    assert torch.equal(two, fused), f"fused redir differs from two launches: {(two - fused).abs().max().item():.3e}"
    close(fused[:, :Co], ref, atol=3e-5, rtol=3e-5, what=f"fused redir [{prec}]")


# ----------------------------------------------------------------------------- IGEV / StereoBase update block (8f #4)
def test_gru_combine_and_sigmoid_tanh_epilogues():
    from openstereo_amd import _lib, ops
    from openstereo_amd.engine import PackedConv3d, ACT_SIGMOID, ACT_TANH
    from openstereo_amd.models.lightstereo import nchw_to_cl, cl_to_nchw
    conv = nn.Conv2d(64, 32, 3, padding=1)
    conv.weight.data = synth_tensor("gru.w", conv.weight.shape, 1) * 3.0
    conv.bias.data = synth_tensor("gru.bias", conv.bias.shape, 1)
    x = T(np.random.default_rng(3).normal(0, 1, (2, 64, 9, 14)).astype(np.float32))
    c = T(np.random.default_rng(4).normal(0, 1, (2, 32, 9, 14)).astype(np.float32))
    with torch.no_grad():
        pre = conv(x) + c
    conv = conv.to(DEV)
    for act, fn in ((ACT_SIGMOID, torch.sigmoid), (ACT_TANH, torch.tanh)):
        y = PackedConv3d(conv, None, act, precision="f32")(nchw_to_cl(x.to(DEV)), residual=nchw_to_cl(c.to(DEV)))
        close(cl_to_nchw(y), fn(pre), atol=2e-5, rtol=2e-5, what=f"act {act}")
    z, q, h = (T(np.random.default_rng(s).uniform(-1, 1, (2, 32, 9, 14)).astype(np.float32)) for s in (5, 6, 7))
    zc, qc, hc = (nchw_to_cl(t.to(DEV)) for t in (z, q, h))
    out = ops.empty_cl(2, 32, 1, 9, 14, DEV)
    _lib.call("osa_gru_combine_f32", zc.data_ptr(), qc.data_ptr(), hc.data_ptr(), out.data_ptr(), 2 * 9 * 14, 32, 32, 32, 32, 32, None, ops._stream())
    close(cl_to_nchw(out), (1 - z) * h + z * q, 0, 0, "gru combine")


@pytest.mark.parametrize("prec", PRECS)
def test_igev_update_block_vs_reference_golden(prec):
    """One full iteration of BasicMultiUpdateBlock (3 ConvGRUs, motion encoder, disp head, mask head) on the
    engine vs the real reference's outputs; then two more iterations vs the oracle (state fed back)."""
    from conftest import igev_update_case
    from oracle import torch_ref as O
    from openstereo_amd import engine
    g = golden("igev_update.npz")
    blk, sd, net, inp, corr, disp = igev_update_case()
    old = engine.get_precision()
    engine.set_precision(prec)
    try:
        blk = blk.to(DEV)
        dv = lambda ts: [t.to(DEV) for t in ts]
        with torch.no_grad():
            n, mask, delta = blk(dv(net), [dv(ts) for ts in inp], corr.to(DEV), disp.to(DEV))
            for k, v in (("net0", n[0]), ("net1", n[1]), ("net2", n[2]), ("mask", mask), ("delta", delta)):
                close(v, g[k], atol=3e-5, rtol=3e-5, what=f"update block {k} [{prec}]")
            # slow-fast schedule of igev_stereo.py:194-201: low-res GRUs only, then a full update
            rn, rm, rd = O.igev_update_block(net, inp, corr, disp, sd)
            rn = O.igev_update_block(rn, inp, None, None, sd, iter16=True, iter08=False, iter04=False, update=False)
            rn, rm, rd = O.igev_update_block(rn, inp, corr, disp + rd, sd)
            n = blk(n, [dv(ts) for ts in inp], iter16=True, iter08=False, iter04=False, update=False)
            n, mask, d2 = blk(n, [dv(ts) for ts in inp], corr.to(DEV), disp.to(DEV) + delta)
    finally:
        engine.set_precision(old)
    close(n[0], rn[0], atol=1e-4, rtol=1e-4, what=f"iteration 2 net0 [{prec}]")
    close(d2, rd, atol=2e-4, rtol=2e-4, what=f"iteration 2 delta [{prec}]")


def test_psmnet_spp_backbone_engine_vs_reference_golden_and_torch_path():
    """PSMNet's SPP feature extractor on the engine (8f #4) vs the real reference's feature map and vs the
    PyTorch-ROCm module path of the same parameters."""
    from openstereo_amd.models.psmnet import PSMNet, _Cfg
    g = golden("psmnet_256x512.npz")
    net = PSMNet(_Cfg(MAX_DISP=64))
    net.load_state_dict(synth_state_dict(net, seed=0, head_gain=3.0), strict=False)
    net = net.to(DEV).eval()
    L, R = synth_images(1, 256, 512, seed=1, max_shift=16.0)
    bb = net.Backbone
    with torch.no_grad():
        assert bb.use_engine
        fe = bb({"left": L.to(DEV), "right": R.to(DEV)})
        bb.use_engine = False
        ft = bb({"left": L.to(DEV), "right": R.to(DEV)})
        bb.use_engine = True
    close(fe["ref_feature"], g["left_feature"], atol=2e-4, rtol=2e-4, what="engine SPP backbone (left)")
    close(fe["tgt_feature"], g["right_feature"], atol=2e-4, rtol=2e-4, what="engine SPP backbone (right)")
    close(fe["ref_feature"], ft["ref_feature"], atol=2e-4, rtol=2e-4, what="engine vs torch SPP backbone")


@pytest.mark.parametrize("prec", PRECS)
def test_gwcnet_batch_invariance_and_odd_size(prec):
    """Batched inference equals per-pair inference (bit for bit in exact-f32 mode: per-batch-item base pointers, 32-bit
    offsets), also for an image size whose quarter resolution is not a multiple of the brick sizes."""
    from openstereo_amd import engine
    from openstereo_amd.models.gwcnet import GwcNet
    old = engine.get_precision()
    engine.set_precision(prec)
    try:
        net = GwcNet()
        net.load_state_dict(synth_state_dict(net, seed=0))
        net = net.to(DEV).eval()
        L, R = synth_images(3, 96, 176, seed=5)              # quarter resolution 24 x 44
        with torch.no_grad():
            both = net({"left": L.to(DEV), "right": R.to(DEV)})["disp_pred"]
            for i in range(3):
                one = net({"left": L[i:i + 1].to(DEV), "right": R[i:i + 1].to(DEV)})["disp_pred"]
                if prec == "f32":
                    assert torch.equal(both[i:i + 1], one), f"pair {i}: batched != single ({(both[i:i+1] - one).abs().max().item():.3e})"
                else:       # f16x3: the operand scale is a power of two derived from the max over the WHOLE batch tensor, so elements below
                    # 2^-18 of that maximum round differently when the batch changes, and the tile / split-K choice (summation order) follows
                    # the pixel count of the batch: agreement to a few 1e-6 relative (disparities ~100 px), not bit for bit
                    assert float((both[i:i + 1] - one).abs().max()) < 5e-4, f"pair {i}: {(both[i:i+1] - one).abs().max().item():.3e}"
        assert torch.isfinite(both).all() and both.std() > 1.0
    finally:
        engine.set_precision(old)


def test_split_activation_format_chain():
    """f16x3 mode: activations handed from one engine layer to the next in the split hi/lo format
    (producer epilogue splits, consumer staging copies): same results as the fp32-tensor chain to fp32
    rounding; with a split residual and a fused redir branch within the usual tolerance of the torch reference."""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d, is_split
    mk = lambda ci, co, k, s, name: (lambda c: (setattr(c.weight, "data", synth_tensor(name, c.weight.shape, 1)), c)[1])(
        nn.Conv3d(ci, co, k, s, k // 2, bias=False))
    ca, cb, cc = mk(64, 32, 3, 1, "sp.a"), mk(32, 32, 3, 1, "sp.b"), mk(32, 64, 3, 2, "sp.c")
    bna, bnb, bnc = _bn_for(32, 2, "sp.a"), _bn_for(32, 3, "sp.b"), _bn_for(64, 4, "sp.c")
    x = T(np.random.default_rng(3).normal(0, 1, (2, 64, 6, 10, 13)).astype(np.float32))
    with torch.no_grad():
        ra = F.relu(bna(ca(x)))
        rb = F.relu(bnb(cb(ra)) + ra)
        rc_ = F.relu(bnc(cc(rb)))
    pa = PackedConv3d(ca.to(DEV), bna.to(DEV), 1, precision="f16x3")
    pb = PackedConv3d(cb.to(DEV), bnb.to(DEV), 1, precision="f16x3")
    pc = PackedConv3d(cc.to(DEV), bnc.to(DEV), 1, precision="f16x3")
    xc = ops.to_cl(x.to(DEV))
    ya_plain = pa(xc)
    ya = pa(xc, out_split=True)
    assert is_split(ya) and not is_split(ya_plain)
    # consumer of a split tensor == consumer of the fp32 tensor (the two carry different power-of-two scales -- bound-based
    # vs measured maximum -- so only elements below 2^-18 of the maximum may round differently)
    close(pb(ya), pb(ya_plain), atol=2e-6, rtol=2e-6, what="split vs fp32 hand-over")
    yb = pb(ya, residual=ya, out_split=True)                 # split input, split residual, split output
    yc = pc(yb)                                              # split input, fp32 output, stride 2
    assert not is_split(yc)
    close(yc[:, :64], rc_, atol=3e-5, rtol=3e-5, what="split chain vs torch")
    with pytest.raises(AssertionError):
        ops.to_ncdhw(yb)                                     # split tensors never leave the engine chain


def test_lightstereo_cost_stage_vs_reference_golden():
    """LightStereo's correlation volume -> 2-D aggregation -> softmax regression on the engine (f16x3) vs the
    output of the reference's own functions (EPE bar 1e-3 px at quarter resolution)."""
    from conftest import lightstereo_stage_case
    from openstereo_amd import engine
    g = golden("lightstereo_stage.npz")
    st, sd, fl, fr0 = lightstereo_stage_case()
    old = engine.get_precision()
    engine.set_precision("f16x3")
    try:
        st = st.to(DEV)
        with torch.no_grad():
            out = st([f.to(DEV) for f in fl], fr0.to(DEV))
    finally:
        engine.set_precision(old)
    close(out["encoding_volume"], g["enc"], atol=2e-4 * max(1.0, float(np.abs(g["enc"]).max())), rtol=1e-4, what="encoding volume")
    epe = np.abs(out["init_disp"].cpu().numpy() - g["init_disp"]).mean()
    assert epe < 1e-3, epe


@pytest.mark.parametrize("prec", PRECS)
def test_igev_refine_loop_vs_reference_golden(prec):
    """BASELINE configs[4]: three GRU refinement iterations (geometry-encoding lookup + update block with the
    slow-fast schedule + disparity update) on the engine vs the reference's own modules."""
    from conftest import igev_refine_case
    from openstereo_amd import engine
    g = golden("igev_refine.npz")
    ref, sd, ml, mr, gvol, net, inp, d0 = igev_refine_case()
    old = engine.get_precision()
    engine.set_precision(prec)
    try:
        ref = ref.to(DEV)
        dv = lambda ts: [t.to(DEV) for t in ts]
        with torch.no_grad():
            out = ref(ml.to(DEV), mr.to(DEV), gvol.to(DEV), dv(net), [dv(ts) for ts in inp], d0.to(DEV), 3)
    finally:
        engine.set_precision(old)
    close(out["disp"], g["disp"], atol=2e-4, rtol=1e-4, what=f"refined disparity [{prec}]")
    close(out["mask_feat_4"], g["mask"], atol=2e-4, rtol=2e-4, what=f"mask features [{prec}]")
    close(out["net_list"][0], g["net0"], atol=1e-4, rtol=1e-4, what=f"hidden state [{prec}]")


# ----------------------------------------------------------------------------- persistent LDS-DMA pipelined conv (PIPE)
PIPE_CASES = [
    # name, Ci, Co, k, pad, dil, (B, D, H, W), residual (None | "split" | "plain"), split output
    ("32-32 k3 ragged", 32, 32, 3, 1, 1, (2, 7, 13, 19), None, True),
    ("32-32 k3 res split", 32, 32, 3, 1, 1, (1, 9, 17, 25), "split", True),
    ("32-32 k3 plain out", 32, 32, 3, 1, 1, (3, 5, 8, 8), None, False),
    ("64-32 k3", 64, 32, 3, 1, 1, (2, 6, 10, 12), None, True),
    ("64-64 k3 res plain", 64, 64, 3, 1, 1, (1, 8, 9, 17), "plain", False),
    ("128-128 k3", 128, 128, 3, 1, 1, (2, 3, 9, 10), None, True),
    ("32-32 1x1x1", 32, 32, 1, 0, 1, (2, 5, 7, 9), None, True),
    ("32-64 many bricks", 32, 64, 3, 1, 1, (1, 13, 40, 33), None, False),
]


@pytest.mark.parametrize("case", PIPE_CASES, ids=[c[0] for c in PIPE_CASES])
def test_pipelined_conv_on_split_inputs_vs_torch(case):
    """Stride-1 f16x3 convs whose input is a split tensor run as persistent workgroups with the brick double-buffered in
    LDS by LDS-DMA (conv_kernel.h, PIPE): ragged volumes (zero fill by the buffer descriptor), several bricks per workgroup,
    several batch items, split / plain residuals and outputs -- against torch, and bit-identical when repeated."""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d, is_split
    name, Ci, Co, k, pad, dil, (B, D, H, W), res_kind, out_split = case
    pre = nn.Conv3d(Ci, Ci, 1, bias=False)                       # produces the split input (identity-like 1x1x1 layer)
    pre.weight.data = torch.eye(Ci).reshape(Ci, Ci, 1, 1, 1).clone()
    conv = nn.Conv3d(Ci, Co, k, 1, pad, dil, bias=False)
    conv.weight.data = synth_tensor(name + ".w", conv.weight.shape, 1)
    bn = _bn_for(Co, 2, name)
    x = T(np.random.default_rng(3).normal(0, 1, (B, Ci, D, H, W)).astype(np.float32))
    res = torch.randn(B, Co, D, H, W, generator=torch.Generator().manual_seed(4)) if res_kind else None
    with torch.no_grad():
        ref = bn(conv(x))
        if res is not None:
            ref = ref + res
        ref = F.relu(ref)
    xs = PackedConv3d(pre.to(DEV), None, 0, precision="f16x3")(ops.to_cl(x.to(DEV)), out_split=True)
    assert is_split(xs)
    rs = None
    if res_kind == "plain":
        rs = ops.to_cl(res.to(DEV))
    elif res_kind == "split":
        eye = nn.Conv3d(Co, Co, 1, bias=False)
        eye.weight.data = torch.eye(Co).reshape(Co, Co, 1, 1, 1).clone()
        rs = PackedConv3d(eye.to(DEV), None, 0, precision="f16x3")(ops.to_cl(res.to(DEV)), out_split=True)
    pc = PackedConv3d(conv.to(DEV), bn.to(DEV), 1, precision="f16x3")
    y = pc(xs, residual=rs, out_split=out_split)
    y2 = pc(xs, residual=rs, out_split=out_split)
    assert torch.equal(y, y2)
    if out_split:                                                # read back through an identity layer
        eye = nn.Conv3d(Co, Co, 1, bias=False)
        eye.weight.data = torch.eye(Co).reshape(Co, Co, 1, 1, 1).clone()
        y = PackedConv3d(eye.to(DEV), None, 0, precision="f16x3")(y)
    close(y[:, :Co], ref, atol=3e-5, rtol=3e-5, what=f"pipelined conv {name}")


MARCH_CASES = [
    # name, Ci, (B, D, H, W), input kind, residual kind, split output, activation
    ("32 fp32 in, ragged", 32, (2, 7, 11, 21), "plain", None, False, "relu"),
    ("64 fp32 in (dres0.0)", 64, (1, 9, 17, 33), "plain", None, True, "relu"),
    ("32 split chain + split res (dres1.2)", 32, (2, 12, 16, 32), "split", "split", True, "none"),
    ("32 split in, plain res, fp32 out (classif.0-like)", 32, (1, 5, 9, 40), "split", "plain", False, "leaky"),
    ("32 many columns, uncut D", 32, (3, 16, 40, 48), "split", None, True, "relu"),
    ("32 D = 3", 32, (1, 3, 8, 16), "plain", None, False, "relu"),
    ("96 channels (3 passes)", 96, (1, 6, 10, 18), "plain", None, False, "relu"),
]


@pytest.mark.parametrize("case", MARCH_CASES, ids=[c[0] for c in MARCH_CASES])
def test_conv3d_marching_form_vs_torch(case, lib):
    """f16x3 3x3x3 stride-1 convolutions with 32 output channels run in the d-marching form (csrc/conv_march.h): pixel columns walked
    along d with three running output planes.  Ragged H / W, cut and uncut D, 2 / 4 / 6 input chunks, fp32 and split inputs, residuals
    and outputs, all three activations -- against torch, against the brick form (exact-f32 mode) and bit-identical when repeated; the
    launch counter proves the form under test is the one that ran."""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d, is_split
    name, Ci, (B, D, H, W), in_kind, res_kind, out_split, act = case
    Co = 32
    eye = lambda c: (lambda m: (setattr(m.weight, "data", torch.eye(c).reshape(c, c, 1, 1, 1).clone()), m)[1])(nn.Conv3d(c, c, 1, bias=False))
    conv = nn.Conv3d(Ci, Co, 3, 1, 1, bias=False)
    conv.weight.data = synth_tensor(name + ".w", conv.weight.shape, 1)
    bn = _bn_for(Co, 2, name)
    x = T(np.random.default_rng(3).normal(0, 1, (B, Ci, D, H, W)).astype(np.float32))
    res = torch.randn(B, Co, D, H, W, generator=torch.Generator().manual_seed(4)) if res_kind else None
    actf = {"relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.01), "none": lambda t: t}[act]
    with torch.no_grad():
        ref = bn(conv(x))
        if res is not None:
            ref = ref + res
        ref = actf(ref)
    xc = ops.to_cl(x.to(DEV))
    xs = PackedConv3d(eye(Ci).to(DEV), None, 0, precision="f16x3")(xc, out_split=True) if in_kind == "split" else xc
    rs = None
    if res_kind == "plain":
        rs = ops.to_cl(res.to(DEV))
    elif res_kind == "split":
        rs = PackedConv3d(eye(Co).to(DEV), None, 0, precision="f16x3")(ops.to_cl(res.to(DEV)), out_split=True)
    acode = {"none": 0, "relu": 1, "leaky": 2}[act]
    pc = PackedConv3d(conv.to(DEV), bn.to(DEV), acode, 0.01, precision="f16x3")
    n0 = lib.osa_conv3d_march_launches()
    y = pc(xs, residual=rs, out_split=out_split)
    assert lib.osa_conv3d_march_launches() == n0 + 1, "the layer did not take the marching form"
    y2 = pc(xs, residual=rs, out_split=out_split)
    assert torch.equal(y, y2)
    assert is_split(y) == out_split
    if out_split:
        y = PackedConv3d(eye(Co).to(DEV), None, 0, precision="f16x3")(y)
    close(y[:, :Co], ref, atol=3e-5, rtol=3e-5, what=f"marching conv {name} vs torch")
    # the brick form (exact-f32 mode never marches) on fp32 tensors
    pb = PackedConv3d(conv.to(DEV), bn.to(DEV), acode, 0.01, precision="f32")
    n1 = lib.osa_conv3d_march_launches()
    yb = pb(xc, residual=None if res is None else ops.to_cl(res.to(DEV)))
    assert lib.osa_conv3d_march_launches() == n1
    close(y[:, :Co], yb[:, :Co], atol=3e-5, rtol=3e-5, what=f"marching conv {name} vs brick form")


def test_dormant_volume_variants():
    """CoExCostVolume / compute_volume / build_sub_volume (cost_volume.py:9-29, 44-56, 108-117) on the engine vs the oracle restatements
    (CoEx also vs the reference's own output), incl. maxdisp > W and a ragged width."""
    from openstereo_amd import ops
    from oracle import torch_ref as R
    g = golden("dormant_volumes.npz")
    x, y = T(g["x"]), T(g["y"])
    for grp in (1, 4):
        got = ops.CoExCostVolume(6, grp)(x.to(DEV), y.to(DEV)).cpu()
        torch.testing.assert_close(got, T(g[f"coex_g{grp}"]), rtol=1e-5, atol=1e-6)
    # the reference's own outputs (its device='cuda' zeros redirected to the CPU when the fixture was generated)
    assert torch.equal(ops.compute_volume(x.to(DEV), y.to(DEV), 7, "left").cpu(), T(g["compute_left"]))
    assert torch.equal(ops.compute_volume(x.to(DEV), y.to(DEV), 7, "right").cpu(), T(g["compute_right"]))
    torch.testing.assert_close(ops.build_sub_volume(x.to(DEV), y.to(DEV), 7).cpu(), T(g["sub_volume"]), rtol=1e-6, atol=1e-6)
    for tag in ("neg", "dil", "negdil"):            # cat_fms with negative start disparities / dilation (psmnet_cost_processor.py:30-47)
        md, st, dil = (int(v) for v in g[f"catfms_{tag}_args"])
        got = ops.cat_fms(x.to(DEV), y.to(DEV), max_disp=md, start_disp=st, dilation=dil)
        assert got.dtype == torch.float32 and torch.equal(got.cpu(), T(g[f"catfms_{tag}"])), tag
    wide = ops.cat_fms(x.to(DEV), y.to(DEV), max_disp=30, start_disp=-25, dilation=1)           # |disparity| >= W: all-zero planes
    assert torch.equal(wide.cpu(), R.cat_fms(x, y, 30, -25, 1))
    for (B, C, H, W, D) in ((2, 12, 5, 37, 9), (1, 8, 3, 6, 10)):
        l, r = rnd((B, C, H, W), 311), rnd((B, C, H, W), 312)
        for side in ("left", "right"):
            assert torch.equal(ops.compute_volume(l.to(DEV), r.to(DEV), D, side).cpu(), R.compute_volume(l, r, D, side)), side
        torch.testing.assert_close(ops.build_sub_volume(l.to(DEV), r.to(DEV), D).cpu(), R.build_sub_volume(l, r, D), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(ops.CoExCostVolume(D - 1, 4)(l.to(DEV), r.to(DEV)).cpu(), R.coex_cost_volume(l, r, D - 1, 4), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(2, 128, 17, 30), (1, 128, 34, 60), (1, 64, 8, 9)])
def test_hidden_state_resampling_kernels(shape):
    """pool2x / interp of the update block (update.py:99-109) on NHWC engine tensors, channel-sliced source and destination, vs torch."""
    import torch.nn.functional as F
    from openstereo_amd import ops
    from openstereo_amd.models.igev_update import _resample_into, pool2x, interp
    from openstereo_amd.models.lightstereo import nchw_to_cl, cl_to_nchw
    B, C, H, W = shape
    x = rnd(shape, 71)
    src = ops.empty_cl(B, C + 64, 1, H, W, DEV)
    src.normal_()
    src[:, :C] = nchw_to_cl(x.to(DEV))[:, :C]
    want_p = F.avg_pool2d(x, 3, stride=2, padding=1)
    dst = ops.empty_cl(B, 32 + C + 12, 1, want_p.shape[2], want_p.shape[3], DEV)
    dst.fill_(7.0)
    _resample_into("pool", src, C, dst, 32)
    close(cl_to_nchw(dst[:, 32:32 + C]), want_p, 1e-6, 1e-6, "pool2x")
    assert float(dst[:, :32].min()) == 7.0 and float(dst[:, 32 + C:].max()) == 7.0          # neighbours untouched
    close(cl_to_nchw(pool2x(nchw_to_cl(x.to(DEV)))), want_p, 1e-6, 1e-6, "pool2x (new tensor)")
    for (Ho, Wo) in ((2 * H, 2 * W), (2 * H - 1, 2 * W + 1), (H, W)):
        want_i = F.interpolate(x, (Ho, Wo), mode="bilinear", align_corners=True)
        dest = ops.empty_cl(B, C, 1, Ho, Wo, DEV)
        close(cl_to_nchw(interp(nchw_to_cl(x.to(DEV)), dest)), want_i, 2e-6, 1e-6, f"interp -> {Ho}x{Wo}")


# ----------------------------------------------------------------------------- B operands through the LDS ring (r4)
BRING_CASES = [
    # name, kind, Ci, Co, k, stride, (D, H, W), out_split
    ("3d 48-32 s1", "conv", 48, 32, 3, 1, (2, 19, 20), False),          # 27 taps (odd run), 3 chunks, ragged bricks (D >= 3 would take the marching form)
    ("3d 64-64 s1 split out", "conv", 64, 64, 3, 1, (5, 9, 17), True),
    ("3d 128-128 s1", "conv", 128, 128, 3, 1, (3, 9, 10), False),       # 2 x 2 waves, 4 N tiles per workgroup: two transfers per wave
    ("3d 32-64 s2", "conv", 32, 64, 3, 2, (8, 12, 20), False),
    ("3d 64-128 s2", "conv", 64, 128, 3, 2, (6, 10, 14), False),
    ("3d 48-24 s2", "conv", 48, 24, 3, 2, (8, 8, 12), False),           # 2-wave tile
    ("1x1x1 64-64", "conv", 64, 64, 1, 1, (3, 4, 7), False),            # a single tap per chunk
    ("2d 64-64", "conv", 64, 64, 3, 1, (1, 40, 50), False),
    ("2d 128-128", "conv", 128, 128, 3, 1, (1, 60, 110), False),        # 32-pixel x 128-channel tile (1 x 4 waves), no split-K at this size
    ("2d 128-128 large map", "conv", 128, 128, 3, 1, (1, 200, 260), False),   # 128 x 128 tile (2 x 2 waves)
    ("2d 32-32", "conv", 32, 32, 3, 1, (1, 33, 47), False),
    ("2d 320-128", "conv", 320, 128, 3, 1, (1, 60, 110), False),         # 20 chunks, several staging passes
    ("2d 32-64 s2", "conv", 32, 64, 3, 2, (1, 30, 44), False),
    ("deconv 64-32", "deconv", 64, 32, 3, 2, (4, 6, 9), False),          # 8 parity classes: runs of 1 / 2 / 4 / 8 taps
    ("deconv 128-64 split out", "deconv", 128, 64, 3, 2, (3, 5, 7), True),
    ("deconv k4 48-24", "deconv4", 48, 24, 4, 2, (3, 4, 6), False),
]


@pytest.mark.parametrize("prec", ["f16x3", "f16"])
@pytest.mark.parametrize("case", BRING_CASES, ids=[c[0] for c in BRING_CASES])
def test_conv_b_ring_is_bit_identical(case, prec):
    """conv_kernel.h BL = 1: weight fragments fetched once per workgroup through the LDS ring instead of once per wave -- the same
    products in the same order, so every output bit equals the per-wave form's (osa_conv_b_ring_mask(0))."""
    from openstereo_amd import _lib, ops
    from openstereo_amd.engine import PackedConv3d
    lib = _lib.load()
    name, kind, Ci, Co, k, s, (D, H, W), out_split = case
    if prec == "f16" and out_split:
        pytest.skip("fp16 outputs are covered by the f16-mode chain tests")
    if kind == "conv":
        m = nn.Conv3d(Ci, Co, (1 if D == 1 else k, k, k), (1 if D == 1 else s, s, s), (0 if D == 1 else k // 2, k // 2, k // 2), bias=False)
    elif kind == "deconv":
        m = nn.ConvTranspose3d(Ci, Co, 3, stride=2, padding=1, output_padding=1, bias=False)
    else:
        m = nn.ConvTranspose3d(Ci, Co, 4, stride=2, padding=1, bias=False)
    m.weight.data = synth_tensor(name + ".w", m.weight.shape, 1)
    bn = _bn_for(Co, 2, name)
    x = ops.to_cl(T(np.random.default_rng(3).normal(0, 1, (2, Ci, D, H, W)).astype(np.float32)).to(DEV))
    pc = PackedConv3d(m.to(DEV), bn.to(DEV), 1, precision=prec)
    kw = {"out_split": True} if out_split else {}
    prev = lib.osa_conv_b_ring_mask(0)
    try:
        n0 = lib.osa_conv_b_ring_launches()
        y_wave = pc(x, **kw).clone()
        assert lib.osa_conv_b_ring_launches() == n0
        lib.osa_conv_b_ring_mask(-1)
        y_ring = pc(x, **kw)
        assert lib.osa_conv_b_ring_launches() == n0 + 1, "the ring form did not run"
    finally:
        lib.osa_conv_b_ring_mask(prev)
    torch.cuda.synchronize()
    assert torch.isfinite(y_ring.float()).all()
    assert torch.equal(y_wave, y_ring), f"{name} [{prec}]: ring form differs by {(y_wave.float() - y_ring.float()).abs().max().item():.3e}"


# ----------------------------------------------------------------------------- d-walking volume builder (r4)
@pytest.mark.parametrize("step", [8, 4])
@pytest.mark.parametrize("case", [
    # name, B, gwc channels, groups, concat channels, H, W, D, mask_left
    ("gwcnet", 2, 320, 40, 12, 9, 96, 48, True),
    ("gwcnet ragged", 1, 320, 40, 12, 5, 75, 21, True),            # W % 32 != 0, D % step != 0
    ("gwcnet D > W", 1, 320, 40, 12, 3, 70, 90, True),              # disparities beyond the map width: zero planes
    ("unmasked concat", 1, 320, 40, 12, 4, 80, 24, False),          # IGEV-style copy (left concat not masked)
    ("K = 4", 1, 64, 16, 8, 6, 140, 17, True),                       # 4 + 4 quads = 8 per voxel: 64-pixel tiles
    ("concat only", 1, 0, 0, 16, 4, 130, 12, True),
], ids=lambda c: c[0])
def test_volume_walking_form_is_bit_identical(case, step):
    """csrc/volume.hip build_volume_walk_kernel (a workgroup walks along d, right window as an LDS ring refilled by a loader wave through
    LDS-DMA): the same lanes run the same fmaf chains and stores as the chunked kernel -- every output bit equal, maximum tracked too."""
    from openstereo_amd import _lib, ops, ranges
    lib = _lib.load()
    name, B, C, G, Cc, H, W, D, mask_left = case
    rng = np.random.default_rng(5)
    gf = ops.empty_cl(2 * B, max(C, 4), 1, H, W, DEV); gf.copy_(T(rng.normal(0, 1, tuple(gf.shape)).astype(np.float32)))
    cf = ops.empty_cl(2 * B, Cc, 1, H, W, DEV); cf.copy_(T(rng.normal(0, 1, tuple(cf.shape)).astype(np.float32)))
    run = lambda: ops.build_cost_volume_from_cl(gf, G, cf, B, D, gwc_channels=C, mask_left=mask_left)
    prev = lib.osa_volume_walk_step(0)
    try:
        n0 = lib.osa_volume_walk_launches()
        v_chunk = run()
        assert lib.osa_volume_walk_launches() == n0
        lib.osa_volume_walk_step(step)
        v_walk = run()
        assert lib.osa_volume_walk_launches() == n0 + 1, "the walking form did not run"
    finally:
        lib.osa_volume_walk_step(prev)
    torch.cuda.synchronize()
    assert torch.equal(v_chunk, v_walk), f"{name}: walking form differs by {(v_chunk - v_walk).abs().max().item():.3e}"
    assert float(ranges.meta_of(v_walk)[::16][:8].max()) == float(v_walk.abs().max()) == float(ranges.meta_of(v_chunk)[::16][:8].max())


@pytest.mark.parametrize("case", [("gwcnet", 2, 320, 40, 12, 9, 96, 48), ("gwcnet ragged", 1, 320, 40, 12, 5, 75, 21), ("concat only", 1, 0, 0, 16, 4, 130, 12)],
                         ids=lambda c: c[0])
def test_volume_written_as_split_tensor(case):
    """osa_build_volume_nhwc_split_f16x3: the fused volume in the f16x3 chain's split format (hi / lo fp16 halves per 16-channel chunk, power-of-
    two scale from the FEATURES' range blocks).  Decoded, it equals the fp32 volume to 2^-21 relative (+ the absolute floor of elements far
    below the bound); the scale is the documented bound; a consumer layer gives the same result from either format."""
    from openstereo_amd import ops, ranges
    from openstereo_amd.engine import PackedConv3d, is_split
    name, B, C, G, Cc, H, W, D = case
    rng = np.random.default_rng(6)
    gf = ops.empty_cl(2 * B, max(C, 4), 1, H, W, DEV); gf.copy_(T(rng.normal(0, 1.5, tuple(gf.shape)).astype(np.float32)))
    cf = ops.empty_cl(2 * B, Cc, 1, H, W, DEV); cf.copy_(T(rng.normal(0, 3.0, tuple(cf.shape)).astype(np.float32)))
    ranges.ensure_meta(gf); ranges.ensure_meta(cf)
    v32 = ops.build_cost_volume_from_cl(gf, G, cf, B, D, gwc_channels=C)
    vs = ops.build_cost_volume_from_cl(gf, G, cf, B, D, gwc_channels=C, out_split=True)
    assert is_split(vs) and not is_split(v32), "the call qualifies for the split form"
    meta = ranges.meta_of(vs)
    scale = float(meta[1])
    bound = max(float(gf.abs().max()) ** 2 if C else 0.0, float(cf.abs().max())) * 1.0625
    assert scale == 2.0 ** (15 - (int(np.floor(np.log2(bound))) + 1)), (scale, bound)          # bound * scale in [2^14, 2^15)
    raw = vs.permute(0, 2, 3, 4, 1).contiguous().view(torch.float16)                           # [B, D, H, W, 2 * VC] halves
    raw = raw.reshape(*raw.shape[:4], -1, 2, 16).float()                                       # [..., chunk, hi | lo, 16]
    dec = ((raw[..., 0, :] + raw[..., 1, :]) / scale).reshape(*raw.shape[:4], -1).permute(0, 4, 1, 2, 3)
    err = (dec - v32).abs()
    tol = v32.abs() * 2.0 ** -21 + 2.0 ** -24 / scale
    assert bool((err <= tol).all()), f"{name}: decoded split volume off by {float((err - tol).max()):.3e}"
    assert float(meta[::16][:8].max()) == float(v32.abs().max())                               # the running maximum tracks the fp32 values
    if (G + 2 * Cc) % 16 == 0 and D >= 3:
        conv = nn.Conv3d(G + 2 * Cc, 32, 3, 1, 1, bias=False)
        conv.weight.data = synth_tensor("vsplit.w", conv.weight.shape, 1)
        pc = PackedConv3d(conv.to(DEV), None, 1, precision="f16x3")
        close(pc(vs), pc(v32), atol=3e-5, rtol=3e-5, what=f"{name}: consumer of the split volume vs of the fp32 volume")
