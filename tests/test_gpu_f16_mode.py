"""The native f16 arithmetic mode (engine precision "f16", csrc PREC_F16): what the reference's AMP configs compute -- cfgs/stereobase/
stereobase_sceneflow.yaml:50, cfgs/lightstereo/lightstereo_s_sceneflow.yaml:36, cfgs/igev/igev_sceneflow_amp.yaml:40 set AMP: true and
trainer_template.py:211,281 wrap every forward in torch.autocast -- fp16 operands, one MFMA per product, fp32 accumulation.

Single layers are checked EXACTLY in the sense that matters: against an fp32 torch convolution of the fp16-ROUNDED operands (products of
fp16 values are exact in fp32, so only the summation order differs), for fp32 and fp16 inputs / residuals / outputs.  Whole modules are
checked the way tests/test_gpu_autocast.py checks the autocast contract: the distance to the fp32 run must stay within twice the distance
the eager torch composition under the same autocast has to ITS fp32 run, and the disparity error against the fp32 result is reported."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import rnd
from openstereo_amd.utils.weights import synth_state_dict, synth_images, synth_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
h = lambda t: t.half().float()          # round to nearest-even fp16


def close(a, b, atol, rtol, what):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} elements off, max err {float(err.max()):.3e}"


def _bn(c, seed, name):
    bn = nn.BatchNorm3d(c)
    bn.load_state_dict({k: synth_tensor(f"{name}.{k}", v.shape, seed) for k, v in bn.state_dict().items()})
    return bn.eval()


CASES = [
    # name, kind, Ci, Co, k, stride, pad, dil, (D,H,W), act, residual
    ("32-32 s1", "conv", 32, 32, 3, 1, 1, 1, (6, 9, 20), "relu", False),
    ("64-32 s1 ragged res", "conv", 64, 32, 3, 1, 1, 1, (5, 7, 11), "relu", True),
    ("32-64 s2", "conv", 32, 64, 3, 2, 1, 1, (8, 12, 20), "relu", False),
    ("64-128 s2 odd", "conv", 64, 128, 3, 2, 1, 1, (6, 10, 14), "relu", False),
    ("128-128 s1", "conv", 128, 128, 3, 1, 1, 1, (3, 6, 10), "relu", False),
    ("1x1 64-64", "conv", 64, 64, 1, 1, 0, 1, (3, 4, 7), "none", True),
    ("24-48 leaky (partial chunk)", "conv", 24, 48, 3, 1, 1, 1, (4, 6, 10), "leaky", False),
    ("8-16", "conv", 8, 16, 3, 1, 1, 1, (5, 6, 7), "relu", False),
    ("48-24 s2", "conv", 48, 24, 3, 2, 1, 1, (8, 8, 12), "leaky", False),
    ("2d 128-128 dil2", "conv", 128, 128, 3, 1, 2, 2, (1, 20, 30), "relu", True),
    ("2d 384-128 (gru)", "conv", 384, 128, 3, 1, 1, 1, (1, 17, 30), "none", False),
    ("deconv k3 64-32", "deconv", 64, 32, 3, 2, 1, 1, (4, 6, 9), "relu", True),
    ("deconv k4 48-24", "deconv", 48, 24, 4, 2, 1, 1, (3, 4, 6), "leaky", False),
]


@pytest.mark.parametrize("in16", [False, True], ids=["fp32-in", "fp16-in"])
@pytest.mark.parametrize("out16", [False, True], ids=["fp32-out", "fp16-out"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_f16_layer_vs_torch_on_rounded_operands(case, in16, out16):
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d
    name, kind, Ci, Co, k, s, p, dil, (D, H, W), act, use_res = case
    if (in16 and Ci % 8) or (out16 and Co % 8):
        pytest.skip("fp16 tensors carry complete 8-channel rows")
    flat = D == 1
    if kind == "deconv":
        conv = nn.ConvTranspose3d(Ci, Co, k, stride=2, padding=1, output_padding=1 if k == 3 else 0, bias=False)
    elif flat:
        conv = nn.Conv2d(Ci, Co, k, s, p, dil, bias=False)
    else:
        conv = nn.Conv3d(Ci, Co, k, s, p, dil, bias=False)
    conv.weight.data = synth_tensor(name + ".w", conv.weight.shape, 1)
    bn = _bn(Co, 2, name)
    x = torch.from_numpy(np.random.default_rng(3).normal(0, 1, (2, Ci, D, H, W)).astype(np.float32))
    refconv = type(conv)(**{kk: getattr(conv, kk) for kk in ("in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation")},
                         **({"output_padding": conv.output_padding} if kind == "deconv" else {}), bias=False)
    refconv.weight.data = h(conv.weight.data)
    with torch.no_grad():
        y0 = refconv(h(x)[:, :, 0] if flat else h(x))
        y0 = y0[:, :, None] if flat else y0
        ref = bn(y0)
        res = torch.randn(ref.shape, generator=torch.Generator().manual_seed(4)) if use_res else None
        r16 = use_res and out16                                   # an fp16 output takes an fp16 residual
        if res is not None:
            ref = ref + (h(res) if r16 else res)
        ref = {"relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.01), "none": lambda t: t}[act](ref)
    bn2 = nn.BatchNorm2d(Co) if flat else bn
    if flat:
        bn2.load_state_dict(bn.state_dict()); bn2.eval()
    pc = PackedConv3d(conv.to(DEV), bn2.to(DEV), {"none": 0, "relu": 1, "leaky": 2}[act], 0.01, precision="f16")
    xc = ops.to_cl(x.to(DEV))
    if in16:
        xc = xc.half()
        assert ops.is_cl(xc)
    rc = None
    if res is not None:
        rc = ops.to_cl(res.to(DEV))
        rc = rc.half() if r16 else rc
    y = pc(xc, residual=rc, out_split=out16)
    assert y.dtype == (torch.float16 if out16 else torch.float32) and ops.is_cl(y)
    # fp32 accumulation of exact products in another order; an fp16 output adds one rounding (2^-11 relative)
    tol = 1.2e-3 if out16 else 3e-5
    close(y[:, :Co], ref, atol=tol * float(ref.abs().max()), rtol=tol, what=f"f16 {name}")


def test_f16_chain_with_fp16_tensors_between_layers():
    """conv a (fp32 in -> fp16 out) -> conv b (fp16 in + fp16 residual -> fp16 out) -> stride-2 conv c (fp16 in -> fp32 out): equals the
    fp32 torch chain on rounded operands with the intermediates rounded to fp16 where the engine stores them as fp16."""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d
    mk = lambda ci, co, k, s, name: (lambda c: (setattr(c.weight, "data", synth_tensor(name, c.weight.shape, 1)), c)[1])(nn.Conv3d(ci, co, k, s, k // 2, bias=False))
    ca, cb, cc = mk(64, 32, 3, 1, "h.a"), mk(32, 32, 3, 1, "h.b"), mk(32, 64, 3, 2, "h.c")
    bna, bnb, bnc = _bn(32, 2, "h.a"), _bn(32, 3, "h.b"), _bn(64, 4, "h.c")
    x = torch.from_numpy(np.random.default_rng(3).normal(0, 1, (2, 64, 6, 10, 13)).astype(np.float32))
    rw = lambda c: F.conv3d
    with torch.no_grad():
        ra = h(F.relu(bna(F.conv3d(h(x), h(ca.weight), None, 1, 1))))
        rb = h(F.relu(bnb(F.conv3d(ra, h(cb.weight), None, 1, 1)) + ra))
        rc_ = F.relu(bnc(F.conv3d(rb, h(cc.weight), None, 2, 1)))
    P = lambda c, b: PackedConv3d(c.to(DEV), b.to(DEV), 1, precision="f16")
    pa, pb, pc = P(ca, bna), P(cb, bnb), P(cc, bnc)
    ya = pa(ops.to_cl(x.to(DEV)), out_split=True)
    yb = pb(ya, residual=ya, out_split=True)
    yc = pc(yb)
    assert ya.dtype == yb.dtype == torch.float16 and yc.dtype == torch.float32
    # a value that sits on an fp16 rounding boundary may round the other way after a different summation order: allow 2 ulp of fp16
    close(ya[:, :32], ra, atol=2e-3 * float(ra.abs().max()), rtol=2e-3, what="chain a")
    close(yc[:, :64], rc_, atol=4e-3 * float(rc_.abs().max()), rtol=4e-3, what="chain c")


def _gwcnet(prec):
    from openstereo_amd import engine
    from openstereo_amd.models.gwcnet import GwcNet
    engine.set_precision(prec)
    net = GwcNet()
    net.load_state_dict(synth_state_dict(net, seed=0))
    return net.to(DEV).eval()


def test_gwcnet_f16_mode_vs_torch_autocast_drift():
    """Whole GwcNet (backbone -> volume -> aggregation -> fused head) in the f16 mode with fp16 tensors between the chained layers: its
    distance to the fp32 result stays within twice the distance of the eager torch composition under torch.autocast(fp16) to ITS fp32
    run (the oracle restatement through PyTorch-ROCm on the same GPU) -- the pin VERDICT r3 asks for -- and the EPE is reported."""
    from openstereo_amd import engine
    from oracle import torch_ref as R
    old = engine.get_precision()
    try:
        L, Rt = synth_images(1, 64, 128, seed=5)
        L, Rt = L.to(DEV), Rt.to(DEV)
        with torch.no_grad():
            net32 = _gwcnet("f32")
            d32 = net32({"left": L, "right": Rt})["disp_pred"].float()
            net16 = _gwcnet("f16")
            d16 = net16({"left": L, "right": Rt})["disp_pred"].float()
            sd = {k: v.to(DEV) for k, v in net32.state_dict().items()}
            e32 = R.gwcnet_forward(L, Rt, sd, 192) if hasattr(R, "gwcnet_forward") else None
            if e32 is not None:
                with torch.autocast("cuda", dtype=torch.float16):
                    e16 = R.gwcnet_forward(L, Rt, sd, 192).float()
        epe = float((d16 - d32).abs().mean())
        print(f"[f16 mode] GwcNet 64x128: EPE vs the fp32 engine run {epe:.4f} px, max {float((d16 - d32).abs().max()):.4f} px")
        assert torch.isfinite(d16).all()
        if e32 is not None:
            drift = float((e16 - e32.float()).abs().mean())
            print(f"[f16 mode] eager torch composition under autocast(fp16): EPE vs its fp32 run {drift:.4f} px")
            assert epe <= 2.0 * drift + 2e-2, f"f16 mode drifts {epe:.4f} px, the autocast composition {drift:.4f} px"
        else:
            assert epe < 0.25
    finally:
        engine.set_precision(old)


def test_autocast_region_selects_the_f16_mode():
    """engine.effective_precision(): layers packed inside a torch.autocast(fp16) region (no_grad) use the f16 mode, outside it -- and
    under bf16 autocast -- the global one; cached packs are keyed on it, so one module serves both."""
    from openstereo_amd import engine
    from openstereo_amd.models.gwcnet import Hourglass
    old = engine.get_precision()
    try:
        engine.set_precision("f16x3")
        hg = Hourglass(8).eval()
        hg.load_state_dict(synth_state_dict(hg, seed=3))
        hg = hg.to(DEV)
        x = rnd((1, 8, 8, 8, 16), 7).to(DEV)
        with torch.no_grad():
            y0 = hg(x)
            assert hg._pack()["c1"].precision == "f16x3"
            with torch.autocast("cuda", dtype=torch.float16):
                y1 = hg(x)
                assert hg._pack()["c1"].precision == "f16"
            with torch.autocast("cuda", dtype=torch.bfloat16):
                hg(x)
                assert hg._pack()["c1"].precision == "f16x3"
            y2 = hg(x)
        assert y1.dtype == torch.float16 and y0.dtype == torch.float32
        assert torch.equal(y0, y2)
        err = float((y1.float() - y0).abs().max()) / float(y0.abs().max())
        assert err < 2e-2, err
    finally:
        engine.set_precision(old)


@pytest.mark.parametrize("prec", ["f32", "f16x3", "f16"])
def test_eight_channel_hourglass_does_not_read_behind_its_input(prec):
    """r6: the fused redir branch of the transposed convolutions loaded whole 16-channel chunks of its input; with 8 channels (StereoBase /
    IGEV hourglasses) the upper half of the LAST voxel's chunk lies behind the tensor.  Zero weights hide finite bytes there, not NaN
    (NaN x 0 = NaN, ReLU(NaN) = 0): the result depended on what the allocator had there before (an order-dependent failure of the test
    above).  Here the bytes behind the input ARE NaN."""
    from openstereo_amd import engine, ops
    from openstereo_amd.models.gwcnet import Hourglass
    old = engine.get_precision()
    try:
        engine.set_precision(prec)
        hg = Hourglass(8).eval()
        hg.load_state_dict(synth_state_dict(hg, seed=3))
        hg = hg.to(DEV)
        x = rnd((1, 8, 8, 8, 16), 7).to(DEV)
        with torch.no_grad():
            xc = ops.to_cl(x)
            n = xc.numel()
            buf = torch.full((n + 4096,), float("nan"), device=DEV)
            buf[:n] = xc.permute(0, 2, 3, 4, 1).reshape(-1)
            xn = buf[:n].view(1, 8, 8, 16, 8).permute(0, 4, 1, 2, 3)            # the same NDHWC tensor with NaN right behind its last voxel
            assert xn.stride() == xc.stride() and torch.equal(xn, xc)
            want = hg.forward_cl(xc).clone()
            got = hg.forward_cl(xn)
        assert torch.isfinite(got).all() and torch.equal(got, want)
        assert float(got[0, :, -1, -1, -1].abs().max()) > 0                     # the last voxel is alive in this fixture
    finally:
        engine.set_precision(old)


@pytest.mark.parametrize("stride", [1, 2])
def test_depthwise_3x3_with_fp16_tensors_vs_the_fp32_kernel(stride):
    """osa_dwconv2d_nhwc_f16io (r6): fp16 in / out tensors of the f16 mode's MobileV2Residual chain -- the fp32 kernel run on the SAME
    fp16-rounded input, its result rounded to fp16: bit-identical (same fmaf order, one rounding at the store)"""
    import torch.nn as nn
    from openstereo_amd import ops
    from openstereo_amd.engine import DepthwiseConv2d, ACT_RELU6
    C, H, W = 192, 23, 37
    conv = nn.Conv2d(C, C, 3, stride, 1, groups=C, bias=False).to(DEV)
    bn = nn.BatchNorm2d(C).to(DEV).eval()
    with torch.no_grad():
        conv.weight.copy_(rnd((C, 1, 3, 3), 1).to(DEV) * 0.3)
        bn.weight.copy_(rnd((C,), 2).abs().to(DEV) + 0.5); bn.bias.copy_(rnd((C,), 3).to(DEV) * 0.1)
        bn.running_mean.copy_(rnd((C,), 4).to(DEV) * 0.1); bn.running_var.copy_(rnd((C,), 5).abs().to(DEV) + 0.5)
    dw = DepthwiseConv2d(conv, bn, ACT_RELU6)
    x32 = ops.to_cl(rnd((2, C, 1, H, W), 6).to(DEV) * 2.0)
    x16 = x32.half()
    assert ops.is_cl(x16)
    want = dw(x16.float())
    got_hh = dw(x16, out_f16=True)
    got_hf = dw(x16)
    got_fh = dw(x16.float(), out_f16=True)
    assert got_hh.dtype == torch.float16 and got_hf.dtype == torch.float32 and got_fh.dtype == torch.float16
    assert torch.equal(got_hf, want) and torch.equal(got_hh, want.half()) and torch.equal(got_fh, want.half())
