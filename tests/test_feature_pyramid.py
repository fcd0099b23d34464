"""Feature pyramids of BASELINE configs[2]-[4] (openstereo_amd/models/feature_pyramid.py) against the reference's OWN classes
(models/stereobase/backbone.py:32-73, models/igev/extractor.py:320-355, models/lightstereo/backbone.py:29-75) built around the same
MobileNetV2-100 trunk mirror (tests/golden/feature_pyramid.npz, make_golden.gen_feature_pyramid).  What this pins: the FPN decoders
(Conv2xUp / Conv2x_IN / FPNLayer, InstanceNorm, replicate-padded out_conv), the forward wiring and the `state_dict` key set.  What it
cannot pin: the trunk against timm (not available offline) -- "parity unpinned" for `conv_stem / bn1 / block*` only."""
import numpy as np
import pytest
import torch

from conftest import golden
from openstereo_amd.utils.weights import synth_state_dict, synth_images

CASES = [("stereobase", "Feature", 61), ("igev", "IGEVFeature", 62), ("lightstereo", "LightStereoBackbone", 63)]


def _build(clsname, seed):
    from openstereo_amd.models import feature_pyramid as FP
    net = getattr(FP, clsname)().eval()
    net.load_state_dict(synth_state_dict(net, seed=seed, gain=0.9))
    return net


@pytest.mark.parametrize("tag,clsname,seed", CASES)
def test_pyramid_keys_and_torch_forward_equal_the_reference_class(tag, clsname, seed):
    """CPU: identical key set (a real checkpoint's `feature.*` / `backbone.*` keys load) and -- same parameters, same image -- the
    reference class's outputs, bit for bit up to fp32 summation order (the torch composition is the reference's op sequence)."""
    g = golden("feature_pyramid.npz")
    net = _build(clsname, seed)
    assert sorted(net.state_dict().keys()) == [str(k) for k in g[f"{tag}_keys"]]
    img, _ = synth_images(2, 128, 256, seed=37)
    with torch.no_grad():
        outs = net(img)
    for i, t in enumerate(outs):
        want = torch.from_numpy(g[f"{tag}_out{i}"])
        assert t.shape == want.shape
        torch.testing.assert_close(t, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))


def test_trunk_mirror_matches_the_timm_architecture_table():
    """mobilenetv2_100 as timm builds it: stem 32, stages (16, 24, 32, 64, 96, 160, 320) x (1, 2, 3, 4, 3, 3, 1) blocks, strides 1/2/2/2/1/2/1,
    expansion 6, timm's parameter names (3.5 M parameters: the published size of the model)."""
    from openstereo_amd.models.feature_pyramid import MobileNetV2Trunk
    t = MobileNetV2Trunk()
    assert [len(s) for s in t.blocks] == [1, 2, 3, 4, 3, 3, 1]
    assert [s[-1].bn2.num_features if i == 0 else s[-1].bn3.num_features for i, s in enumerate(t.blocks)] == [16, 24, 32, 64, 96, 160, 320]
    keys = set(t.state_dict().keys())
    assert {"conv_stem.weight", "bn1.running_var", "blocks.0.0.conv_dw.weight", "blocks.0.0.conv_pw.weight", "blocks.1.0.conv_pw.weight",
            "blocks.1.0.conv_dw.weight", "blocks.1.1.conv_pwl.weight", "blocks.6.0.bn3.bias"} <= keys
    n = sum(p.numel() for p in t.parameters())
    assert 1.8e6 < n < 1.9e6, n            # features-only trunk: 3.5 M of the classifier model minus conv_head (0.41 M) and the 1.28 M classifier
    x = torch.zeros(1, 3, 64, 128)
    y = t.act1(t.bn1(t.conv_stem(x)))
    shapes = []
    for s in t.blocks:
        y = s(y)
        shapes.append(tuple(y.shape[1:]))
    assert shapes == [(16, 32, 64), (24, 16, 32), (32, 8, 16), (64, 4, 8), (96, 4, 8), (160, 2, 4), (320, 2, 4)]


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("tag,clsname,seed", CASES)
def test_pyramid_engine_path_matches_the_reference_class(tag, clsname, seed, prec):
    """GPU, eval: the whole pyramid on the engine (NHWC; fused conv + BN + ReLU6, depthwise kernel, InstanceNorm kernel, channel-slice
    concat) against the reference class's outputs."""
    from openstereo_amd import engine
    g = golden("feature_pyramid.npz")
    old = engine.get_precision()
    try:
        engine.set_precision(prec)
        net = _build(clsname, seed).cuda()
        img, _ = synth_images(2, 128, 256, seed=37)
        with torch.no_grad():
            assert net._engine_ok(img.cuda())
            outs = net(img.cuda())
        for i, t in enumerate(outs):
            want = torch.from_numpy(g[f"{tag}_out{i}"])
            err = float((t.cpu() - want).abs().max()) / float(want.abs().max())
            assert err < 1e-4, (tag, i, err)
    finally:
        engine.set_precision(old)


@pytest.mark.gpu
def test_instance_norm_kernel_vs_torch():
    """csrc/norm.hip alone: ragged channel count (padded quad), large mean (shifted-data statistics), channel-slice output, both activations."""
    import torch.nn.functional as F
    from openstereo_amd import ops
    from openstereo_amd.models.feature_pyramid import instance_norm_act_cl
    from openstereo_amd.models.lightstereo import nchw_to_cl
    g = torch.Generator().manual_seed(5)
    for (B, C, H, W), act, slope in (((2, 48, 33, 70), 2, 0.01), ((1, 22, 8, 8), 1, 0.0), ((3, 96, 136, 240), 0, 0.0)):
        x = torch.randn(B, C, H, W, generator=g) * 3.0 + 50.0 * torch.randn(1, C, 1, 1, generator=g)
        want = F.instance_norm(x)
        want = F.leaky_relu(want, slope) if act == 2 else (F.relu(want) if act == 1 else want)
        xc = nchw_to_cl(x.cuda())
        out = ops.empty_cl(B, 128, 1, H, W, "cuda")
        out.zero_()
        y = instance_norm_act_cl(xc, C, act, slope, out=out, out_off=8)
        got = y[:, 8:8 + C, 0].cpu()
        torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)
        assert float(y[:, :8].abs().max()) == 0.0 and float(y[:, 8 + (C + 3) // 4 * 4:].abs().max()) == 0.0
