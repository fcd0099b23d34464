"""GPU: the f16x3 (split-precision) conv mode stays at fp32-class accuracy whatever the magnitude of its
operands.  Activations carry a device-side range block (max |value|); every kernel
scales its operands by a power of two derived from it, nothing is clamped.  The sweeps below scale the
inputs by 2^-20 ... 2^14, plant an activation beyond the fp16 range, and feed gradient-sized values, and
hold the SAME tolerances as the unit-scale tests (2e-5 abs at unit scale + 2e-5 rel; EPE < 1e-3)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from test_gpu_parity import close, g2, T, DEV, _bn_for
from openstereo_amd.utils.weights import synth_state_dict, synth_images, synth_tensor

pytestmark = pytest.mark.gpu

SCALES = [2.0 ** -20, 2.0 ** -12, 2.0 ** -6, 1.0, 2.0 ** 6, 2.0 ** 12, 2.0 ** 14]


def _conv(ci, co, k, name, stride=1):
    c = nn.Conv3d(ci, co, k, stride, k // 2, bias=False)
    c.weight.data = synth_tensor(name, c.weight.shape, 1)
    return c


@pytest.mark.parametrize("scale", SCALES, ids=lambda s: f"2^{int(np.log2(s))}")
def test_plain_conv_is_scale_invariant(scale):
    """conv(x * s) == s * conv(x) to fp32 rounding: no BN, no activation, so every bit of the operand range
    shows up in the output.  Tolerance scales with s (2e-5 at unit scale)."""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d
    conv = _conv(32, 32, 3, "rng.a")
    x = T(np.random.default_rng(3).normal(0, 1, (2, 32, 6, 9, 13)).astype(np.float32)) * scale
    with torch.no_grad():
        ref = conv(x)
    y = PackedConv3d(conv.to(DEV), None, 0, precision="f16x3")(ops.to_cl(x.to(DEV)))
    close(y[:, :32], ref, atol=2e-5 * scale, rtol=2e-5, what=f"f16x3 conv at input scale {scale:g}")


@pytest.mark.parametrize("scale", SCALES, ids=lambda s: f"2^{int(np.log2(s))}")
def test_split_chain_is_scale_invariant(scale):
    """Two layers handing a SPLIT tensor over (producer picks the scale of the hi/lo halves from its output
    bound, consumer undoes it), plus a split residual: linear chain, so the result scales with the input."""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d
    ca, cb = _conv(32, 32, 3, "rng.b"), _conv(32, 32, 3, "rng.c")
    x = T(np.random.default_rng(4).normal(0, 1, (1, 32, 5, 10, 12)).astype(np.float32)) * scale
    with torch.no_grad():
        ya = ca(x)
        ref = cb(ya) + ya
    pa = PackedConv3d(ca.to(DEV), None, 0, precision="f16x3")
    pb = PackedConv3d(cb.to(DEV), None, 0, precision="f16x3")
    t = pa(ops.to_cl(x.to(DEV)), out_split=True)
    y = pb(t, residual=t)
    close(y[:, :32], ref, atol=6e-5 * scale, rtol=3e-5, what=f"split chain at input scale {scale:g}")


def test_activation_beyond_fp16_range_is_not_clamped():
    """One activation of 3e5 (> 65504, the fp16 maximum) among N(0,1) values: round 1 saturated it silently
    (35 % error in its 27-voxel neighbourhood); now the tensor is scaled by 2^-4 and everything is exact to
    fp32 rounding."""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d
    conv = _conv(32, 32, 3, "rng.d")
    x = T(np.random.default_rng(5).normal(0, 1, (1, 32, 6, 8, 9)).astype(np.float32))
    x[0, 7, 3, 4, 5] = 3.0e5
    with torch.no_grad():
        ref = conv(x)
    assert float(ref.abs().max()) > 1e4
    for split in (False, True):
        pc = PackedConv3d(conv.to(DEV), None, 0, precision="f16x3")
        y = pc(ops.to_cl(x.to(DEV)), out_split=split)
        if split:                                      # read the split tensor back through a 1x1x1 identity layer
            eye = nn.Conv3d(32, 32, 1, bias=False)
            eye.weight.data = torch.eye(32).reshape(32, 32, 1, 1, 1).clone()
            y = PackedConv3d(eye.to(DEV), None, 0, precision="f16x3")(y)
        close(y[:, :32], ref, atol=2e-5, rtol=2e-5, what=f"outlier activation (split={split})")


def test_inf_input_is_loud_not_clamped():
    """A non-finite activation must poison the outputs it touches (as in fp32), never come out as a finite number."""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d
    conv = _conv(32, 32, 3, "rng.e")
    x = torch.zeros(1, 32, 4, 8, 8)
    x[0, 0, 2, 4, 4] = float("inf")
    y = PackedConv3d(conv.to(DEV), None, 0, precision="f16x3")(ops.to_cl(x.to(DEV)))
    assert not torch.isfinite(y[0, :32, 2, 4, 4]).all()


@pytest.mark.parametrize("scale", [2.0 ** -12, 2.0 ** -6, 2.0 ** 6, 2.0 ** 12], ids=lambda s: f"2^{int(np.log2(s))}")
def test_gwc_disp_processor_f16x3_volume_scale_sweep(scale):
    """GwcDispProcessor (dres0..classif3 on split tensors, fused redir branches, fused upsample + soft-argmin)
    in f16x3 on a volume scaled by 2^-12 ... 2^12, vs the CPU oracle on the same scaled volume."""
    from conftest import golden
    from oracle import torch_ref as O
    from openstereo_amd import engine, ops
    from openstereo_amd.models.gwcnet import GwcDispProcessor
    g = golden("gwc_disp.npz")
    vol = T(g["volume"]) * scale
    dp = GwcDispProcessor(maxdisp=32)
    sd = synth_state_dict(dp, seed=4)
    dp.load_state_dict(sd)
    with torch.no_grad():
        taps = {}
        cost3_ref = O.gwc_aggregate(vol, {"DispProcessor." + k: v for k, v in sd.items()}, taps=taps)
        disp_ref = O.upsample_regression(cost3_ref, 32, 32, 64)
    old = engine.get_precision()
    engine.set_precision("f16x3")
    try:
        dp = dp.to(DEV).eval()
        with torch.no_grad():
            cost3 = dp.aggregate_cl(ops.to_cl(vol.to(DEV)))
            disp = dp({"cost_volume": vol.to(DEV), "left": torch.zeros(1, 3, 32, 64, device=DEV)})["inference_disp"]["disp_est"]
    finally:
        engine.set_precision(old)
    mag = max(1.0, float(cost3_ref.abs().max()))
    close(cost3, cost3_ref, atol=5e-4 * mag, rtol=1e-4, what=f"cost3 at volume scale {scale:g}")
    epe = float((disp.cpu() - disp_ref).abs().mean())
    assert epe < 1e-3, f"EPE {epe} at volume scale {scale:g}"


@pytest.mark.parametrize("scale", [2.0 ** -6, 2.0 ** 6], ids=lambda s: f"2^{int(np.log2(s))}")
def test_gwcnet_f16x3_image_scale_sweep(scale):
    """Whole GwcNet (engine backbone on split tensors, volume, aggregation, head) in f16x3 with the input images
    scaled by 2^-6 / 2^6 -- every intermediate range moves -- vs the CPU oracle: EPE < 1e-3."""
    from oracle import torch_ref as O
    from openstereo_amd import engine
    from openstereo_amd.models.gwcnet import GwcNet
    net = GwcNet()
    sd = synth_state_dict(net, seed=0)
    net.load_state_dict(sd)
    L, R = synth_images(1, 64, 128, seed=1)
    L, R = L * scale, R * scale
    with torch.no_grad():
        ref = O.gwcnet_forward(L, R, sd)
    old = engine.get_precision()
    engine.set_precision("f16x3")
    try:
        net = net.to(DEV).eval()
        with torch.no_grad():
            disp = net({"left": L.to(DEV), "right": R.to(DEV)})["disp_pred"]
    finally:
        engine.set_precision(old)
    epe = float((disp.cpu() - ref).abs().mean())
    assert epe < 1e-3, f"EPE {epe} at image scale {scale:g}"


@pytest.mark.parametrize("shape", [(1, 32, 4, 9, 11), (2, 3, 5, 7), (1, 7), (3, 64, 33, 65)])
def test_input_meta_of_outside_tensors_is_the_exact_maximum(shape):
    """Operands that reach an f16x3 layer from torch ops get their range from torch's infinity norm or -- ranges.ENGINE_AMAX -- from
    osa_amax_f32 (dense fp32, 16-byte aligned: contiguous, channels-last, permuted; anything else falls back): the exact max |x| either way."""
    from openstereo_amd import ranges
    from openstereo_amd.ranges import amax_of, input_meta
    g = torch.Generator(device="cpu").manual_seed(5)
    x = (torch.randn(*shape, generator=g) * 3).to(DEV)
    variants = [x, x.to(torch.float16), x[..., 1:], x.flatten()[1:]]
    if x.dim() == 4:
        variants += [x.contiguous(memory_format=torch.channels_last), x.permute(0, 2, 3, 1)]
    if x.dim() == 5:
        variants += [x.contiguous(memory_format=torch.channels_last_3d)]
    old = ranges.ENGINE_AMAX
    try:
        for use_kernel in (False, True):
            ranges.ENGINE_AMAX = use_kernel
            for v in variants:
                got = float(amax_of(input_meta(v)))
                assert got == float(v.float().abs().max()), (use_kernel, tuple(v.shape), v.stride(), v.dtype)
    finally:
        ranges.ENGINE_AMAX = old


def test_range_block_tracks_max_and_graph_replay_is_reproducible():
    """The producing kernel folds max |y| into the output's range block; a captured graph re-zeroes its own blocks,
    so replays are bit-identical to the eager result even when an earlier replay saw larger values."""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d, meta_of
    from openstereo_amd.ranges import amax_of
    conv = _conv(32, 32, 3, "rng.f")
    pc = PackedConv3d(conv.to(DEV), None, 1, precision="f16x3")
    pc2 = PackedConv3d(_conv(32, 32, 3, "rng.g").to(DEV), None, 0, precision="f16x3")
    x = torch.randn(1, 32, 4, 9, 11, device=DEV)
    xc = ops.to_cl(x)
    y = pc(xc, out_split=True)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = F.relu(conv.to(DEV)(x))
    assert abs(float(amax_of(meta_of(y))) - float(ref.abs().max())) <= 1e-4 * float(ref.abs().max())
    s = float(meta_of(y)[1])
    assert s > 0 and np.log2(s) == int(np.log2(s))                      # a power of two
    eager = pc2(y).clone()
    static_x = ops.to_cl(x.clone())
    for _ in range(2):
        pc2(pc(static_x, out_split=True))
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = pc2(pc(static_x, out_split=True))
    static_x.mul_(64.0)                                                  # a replay with much larger values ...
    graph.replay()
    static_x.mul_(1.0 / 64.0)                                            # ... must not leak into the next one
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


def test_gradient_sized_operands_in_dgrad():
    """data gradients of 1e-6 .. 1e-8 through the f16x3 dgrad path keep fp32-class accuracy."""
    from openstereo_amd import autograd as AG
    conv = _conv(32, 32, 3, "rng.h")
    x = T(np.random.default_rng(6).normal(0, 1, (1, 32, 5, 8, 10)).astype(np.float32))
    dy = T(np.random.default_rng(7).normal(0, 1, (1, 32, 5, 8, 10)).astype(np.float32)) * 1e-7
    xr = x.clone().requires_grad_(True)
    F.conv3d(xr, conv.weight, None, 1, 1).backward(dy)
    xg = x.to(DEV).requires_grad_(True)
    y = AG.conv3d(xg, conv.weight.detach().to(DEV), None, 1, 1, 1, precision="f16x3")
    y.backward(dy.to(DEV))
    close(xg.grad, xr.grad, atol=2e-5 * 1e-7, rtol=3e-5, what="f16x3 dgrad with 1e-7 gradients")
