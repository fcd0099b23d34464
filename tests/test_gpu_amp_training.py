"""Training at the reference's own arithmetic (VERDICT r4 missing #2 / next #3a).

The reference trains StereoBase / LightStereo / IGEV under `torch.autocast` + `GradScaler` (stereo/modeling/trainer_template.py:211,217-226;
cfgs/stereobase/stereobase_sceneflow.yaml:50): every convolution, its data gradient and its weight gradient multiply fp16 operands and
accumulate in fp32.  Since r5 the differentiable engine convolutions (openstereo_amd/autograd.py) do the same inside an fp16 autocast
region -- the native "f16" mode: operands rounded to fp16 (nearest even) when they are staged, ONE MFMA per product, no range reductions --
instead of spending the f16x3 mode's three MFMAs on accuracy autocast has given up.

Pinned here: (1) forward / data gradient / weight gradient of every layer kind against torch run on the SAME fp16-rounded operands in
fp32 (the products of fp16 numbers are exact in fp32, so only the summation order differs); (2) an autocast region with grad enabled
selects that mode; (3) a whole AMP training step (StereoBase cost stage: the configs[2] hot path) drifts from the fp32 step no more than
twice what the PyTorch-ROCm eager composition under the same autocast does."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from openstereo_amd.utils.weights import synth_state_dict, synth_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rn(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).normal(0, 1, shape) * scale).astype(np.float32))


def r16(t):
    """what the kernels do to an operand: round to fp16 (nearest even), compute with that value"""
    return t.half().float()


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))


LAYERS = [  # name, kind, Ci, Co, k, stride, pad, opad, dims
    ("conv 32-32 s1", "conv3d", 32, 32, 3, 1, 1, 0, (6, 9, 12)),
    ("conv 64-40 s1 ragged", "conv3d", 64, 40, 3, 1, 1, 0, (5, 7, 11)),
    ("conv 32-64 s2", "conv3d", 32, 64, 3, 2, 1, 0, (8, 12, 16)),
    ("conv 1x1x1 48-32", "conv3d", 48, 32, 1, 1, 0, 0, (3, 9, 10)),
    ("conv2d 3x3 384-128 (gru)", "conv2d", 384, 128, 3, 1, 1, 0, (1, 20, 46)),
    ("conv2d 3x3 36-7 ragged", "conv2d", 36, 7, 3, 1, 1, 0, (1, 9, 33)),
    ("deconv3d k3 64-32", "deconv3d", 64, 32, 3, 2, 1, 1, (3, 5, 7)),
    ("deconv3d k4 48-24", "deconv3d", 48, 24, 4, 2, 1, 0, (4, 5, 9)),
    ("deconv2d k4 64-9", "deconv2d", 64, 9, 4, 2, 1, 0, (1, 20, 46)),
]


def _fns(kind, s, p, op):
    from openstereo_amd import autograd as AG
    if kind == "conv3d":
        return (lambda x, w: F.conv3d(x, w, None, s, p)), (lambda x, w, prec: AG.conv3d(x, w, None, s, p, 1, precision=prec)), False
    if kind == "conv2d":
        return (lambda x, w: F.conv2d(x, w, None, s, p)), (lambda x, w, prec: AG.conv2d(x, w, None, s, p, 1, precision=prec)), False
    if kind == "deconv3d":
        return (lambda x, w: F.conv_transpose3d(x, w, None, 2, p, op)), (lambda x, w, prec: AG.conv_transpose3d(x, w, None, 2, p, op, precision=prec)), True
    return (lambda x, w: F.conv_transpose2d(x, w, None, 2, p, op)), (lambda x, w, prec: AG.conv_transpose2d(x, w, None, 2, p, op, precision=prec)), True


@pytest.mark.parametrize("case", LAYERS, ids=[c[0] for c in LAYERS])
def test_f16_training_kernels_vs_torch_on_rounded_operands(case):
    """forward, dx and dW of the native f16 mode = torch (fp32) on the fp16-rounded x, w and dy: fp32 accumulation of exact products, so
    the two agree to summation-order rounding; and deterministic (two runs bit-identical).  The gradient dy has GradScaler-sized magnitude."""
    name, kind, Ci, Co, k, s, p, op, (D, H, W) = case
    ref, eng, transposed = _fns(kind, s, p, op)
    two_d = kind.endswith("2d")
    wshape = ((Ci, Co) if transposed else (Co, Ci)) + ((k, k) if two_d else (k, k, k))
    w = (synth_tensor(name + ".w", wshape, 1) * 3.0).to(DEV)
    x = (rn((2, Ci, H, W) if two_d else (2, Ci, D, H, W), 21) * 2.0).to(DEV)
    xr, wr = r16(x).requires_grad_(), r16(w).requires_grad_()
    y = ref(xr, wr)
    gy = (torch.randn(y.shape, generator=torch.Generator().manual_seed(5)) * 7.0).to(DEV)     # "scaled" gradients: O(10), far from fp16's edges
    y.backward(r16(gy))
    outs = []
    for _ in range(2):
        xe, we = x.clone().requires_grad_(), w.clone().requires_grad_()
        ye = eng(xe, we, "f16")
        ye.backward(gy)
        outs.append((ye.detach(), xe.grad, we.grad))
    for a, b in zip(*outs):
        assert torch.equal(a, b), "deterministic"
    ye, dx, dw = outs[0]
    assert rel(ye, y) < 2e-6, ("fwd", rel(ye, y))
    assert rel(dx, xr.grad) < 2e-6, ("dx", rel(dx, xr.grad))
    assert rel(dw, wr.grad) < 5e-6, ("dw", rel(dw, wr.grad))
    # and it IS a different arithmetic from the fp32-class modes: the distance to the unrounded fp32 result is fp16-sized
    xf, wf = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref(xf, wf).backward(gy)
    assert 1e-5 < rel(dw, wf.grad) < 5e-3


def test_autocast_region_with_grad_selects_the_native_f16_mode():
    """engine.train_precision: inside torch.autocast(fp16) the differentiable convolutions run the f16 mode (bit-identical to an explicit
    precision="f16" call), outside -- and inside a bf16 region -- the global mode; the results come back in the dtype the reference's
    convolution would return there."""
    from openstereo_amd import autograd as AG, engine
    x = (rn((1, 32, 4, 9, 12), 3) * 2.0).to(DEV)
    w = (synth_tensor("sel.w", (32, 32, 3, 3, 3), 1) * 3.0).to(DEV).requires_grad_()
    old = engine.get_precision()
    engine.set_precision("f16x3")
    try:
        assert engine.train_precision() == "f16x3"
        explicit16 = AG.conv3d(x, w, None, 1, 1, 1, precision="f16")
        x3 = AG.conv3d(x, w, None, 1, 1, 1)
        with torch.autocast("cuda", dtype=torch.float16):
            assert engine.train_precision() == "f16" and torch.is_grad_enabled()
            inside = AG.conv3d(x, w, None, 1, 1, 1)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert engine.train_precision() == "f16x3"
            inside_bf = AG.conv3d(x, w, None, 1, 1, 1)
        assert torch.equal(inside.float(), explicit16.float().to(inside.dtype).float())
        assert not torch.equal(explicit16, x3) and rel(explicit16, x3) < 2e-3
        assert torch.equal(inside_bf.float(), x3.to(inside_bf.dtype).float())
    finally:
        engine.set_precision(old)


def test_stereobase_cost_stage_amp_step_drifts_no_more_than_the_eager_autocast_composition():
    """configs[2] hot path, one AMP training step (autocast + GradScaler, frozen BN): gradients of the engine's native f16 path vs the fp32
    step, against the same distance for the PyTorch-ROCm eager composition (stock conv / conv_transpose kernels) under the same autocast."""
    from openstereo_amd import autograd as AG, engine
    from openstereo_amd.models.igev_style import StereoBaseCostStage

    def build():
        st = StereoBaseCostStage(max_disp=64, num_groups=8, concat_channels=8, backbone_channels=[48, 64, 192, 120])
        st.load_state_dict(synth_state_dict(st, seed=8, head_gain=20.0))
        st = st.to(DEV).train()
        for m in st.modules():
            if isinstance(m, nn.modules.batchnorm._BatchNorm):
                m.eval()
        return st
    g = torch.Generator().manual_seed(80)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    B, H, W = 1, 32, 64
    x = [r(B, 96, H, W), r(B, 96, H, W), r(B, 8, H, W), r(B, 8, H, W)]
    feats = [None, r(B, 64, H // 2, W // 2), r(B, 192, H // 4, W // 4), r(B, 120, H // 8, W // 8)]
    gt = torch.rand(B, 1, H, W, generator=g).to(DEV) * 12

    def step(st, amp, eager=False):
        scaler = torch.amp.GradScaler("cuda", enabled=amp, init_scale=1024.0)
        opt = torch.optim.SGD(st.parameters(), lr=0.0)
        ctx = _stock_torch_convs() if eager else _null()
        with ctx:
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                out = st(*x, feats)
                loss = F.smooth_l1_loss(out["init_disp"].float(), gt)
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
        gr = {k: p.grad.detach().float().clone() for k, p in st.named_parameters() if p.grad is not None}
        assert all(torch.isfinite(v).all() for v in gr.values()), "scaled fp16 gradients overflowed at init_scale 1024"
        return float(loss), gr

    old = engine.get_precision()
    engine.set_precision("f16x3")
    try:
        l32, g32 = step(build(), False)
        l16, g16 = step(build(), True)
        le, ge = step(build(), True, eager=True)
    finally:
        engine.set_precision(old)
    assert abs(l16 - l32) < 2e-2 * abs(l32) and len(g32) > 20 and g16.keys() == g32.keys() == ge.keys()
    worst = 0.0
    for k in g32:
        s = float(g32[k].abs().max()) + 1e-20
        e_eng, e_ref = float((g16[k] - g32[k]).abs().max()) / s, float((ge[k] - g32[k]).abs().max()) / s
        worst = max(worst, e_eng)
        assert e_eng <= 2.0 * e_ref + 2e-2, (k, e_eng, e_ref)
    print(f"[amp step] worst relative gradient distance to the fp32 step: {worst:.3e}")
    assert worst > 1e-5, "the AMP step must have run the fp16 arithmetic (distance to fp32 is fp16-sized, not zero)"


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _stock_torch_convs:
    """run the model's convolutions on stock PyTorch-ROCm kernels: the engine's differentiable conv entry points replaced by torch's"""

    def __enter__(self):
        from openstereo_amd import autograd as AG
        self.AG, self.saved = AG, {n: getattr(AG, n) for n in ("conv3d", "conv_transpose3d", "conv2d", "conv_transpose2d")}
        AG.conv3d = lambda x, w, b=None, stride=1, padding=0, dilation=1, precision=None: F.conv3d(x, w, b, stride, padding, dilation)
        AG.conv2d = lambda x, w, b=None, stride=1, padding=0, dilation=1, precision=None: F.conv2d(x, w, b, stride, padding, dilation)
        AG.conv_transpose3d = lambda x, w, b=None, stride=2, padding=1, output_padding=0, precision=None: F.conv_transpose3d(x, w, b, stride, padding, output_padding)
        AG.conv_transpose2d = lambda x, w, b=None, stride=2, padding=1, output_padding=0, precision=None: F.conv_transpose2d(x, w, b, stride, padding, output_padding)
        return self

    def __exit__(self, *a):
        for n, f in self.saved.items():
            setattr(self.AG, n, f)
        return False


F16IO = [  # name, kind, Ci, Co, k, pad, dims, input layout
    ("conv3d 32-32 ncdhw", "conv3d", 32, 32, 3, 1, (6, 9, 12), "nchw"),
    ("conv3d 64-40 cl", "conv3d", 64, 40, 3, 1, (5, 7, 11), "cl"),
    ("conv3d 1x1x1 48-32 cl", "conv3d", 48, 32, 1, 0, (3, 9, 10), "cl"),
    ("conv2d 3x3 384-128 cl (gru)", "conv2d", 384, 128, 3, 1, (1, 20, 46), "cl"),
    ("conv2d 3x3 64-7 nchw (fp32 result)", "conv2d", 64, 7, 3, 1, (1, 9, 33), "nchw"),
    ("conv2d slice of a wider map", "conv2d", 32, 32, 3, 1, (1, 12, 20), "slice"),
]


@pytest.mark.parametrize("case", F16IO, ids=[c[0] for c in F16IO])
def test_fp16_tensor_path_vs_torch_on_the_same_fp16_tensors(case):
    """_Conv3dF16IO: fp16 tensors in (NCHW, channels-last, or a channel slice of a wider NHWC map), fp16 result / data gradient, fp32 weight
    gradient read from the fp16 activation and gradient -- against torch (fp32 arithmetic) on the SAME fp16 values; selected inside an fp16
    autocast region for fp16 inputs; the saved activation is the fp16 tensor."""
    from openstereo_amd import autograd as AG
    name, kind, Ci, Co, k, p, (D, H, W), layout = case
    two_d = kind == "conv2d"
    w = (synth_tensor(name + ".w", (Co, Ci) + ((k, k) if two_d else (k, k, k)), 1) * 3.0).to(DEV)
    bias = (rn((Co,), 3) * 0.5).to(DEV)
    x32 = (rn((2, Ci, H, W) if two_d else (2, Ci, D, H, W), 21) * 2.0).to(DEV)
    x16 = x32.half()
    if layout == "cl":
        x16 = x16.contiguous(memory_format=torch.channels_last if two_d else torch.channels_last_3d)
    elif layout == "slice":
        wide = torch.cat([x16, torch.zeros_like(x16)], 1).contiguous(memory_format=torch.channels_last)
        x16 = wide[:, :Ci]
    ref = (lambda a, b, c: F.conv2d(a, b, c, 1, p)) if two_d else (lambda a, b, c: F.conv3d(a, b, c, 1, p))
    xr, wr, br = x16.float().detach().requires_grad_(), r16(w).requires_grad_(), bias.clone().requires_grad_()
    y = ref(xr, wr, br)
    gy = (torch.randn(y.shape, generator=torch.Generator().manual_seed(5)) * 7.0).to(DEV).half()
    y.backward(gy.float())
    xe, we, be = x16.detach().requires_grad_(), w.clone().requires_grad_(), bias.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.float16):
        ye = (AG.conv2d if two_d else AG.conv3d)(xe, we, be, 1, p, 1)
    assert ye.dtype == (torch.float16 if Co % 8 == 0 else torch.float32), ye.dtype
    assert type(ye.grad_fn).__name__.startswith("_Conv3dF16IO"), type(ye.grad_fn).__name__
    ye.backward(gy if ye.dtype == torch.float16 else gy.float())
    assert xe.grad.dtype == torch.float16 and we.grad.dtype == torch.float32
    # results are rounded to fp16 once (2^-11 relative); the weight gradient is fp32 sums of exact products
    assert rel(ye, y) < 1e-3, ("fwd", rel(ye, y))
    assert rel(xe.grad, xr.grad) < 1e-3, ("dx", rel(xe.grad, xr.grad))
    assert rel(we.grad, wr.grad) < 5e-6, ("dw", rel(we.grad, wr.grad))
    assert rel(be.grad, br.grad) < 1e-5, ("db", rel(be.grad, br.grad))


def test_stereobase_whole_model_amp_step_at_size_drifts_no_more_than_the_eager_autocast_composition():
    """BASELINE configs[2] as `bench.py --workload stereobase_e2e_train --amp` times it: the WHOLE model at the SceneFlow
    training crop 320x736, 22 GRU iterations, one autocast + GradScaler step (trainer_template.py:211-226) with frozen BatchNorm.  The fp32-class
    step of the same model is pinned to the reference's CPU autograd (test_gpu_models_e2e.py::test_stereobase_training_step_at_size_...);
    here the native-f16 engine step -- batched multi-tile weight gradients, channel sums, accumulating lookup gradient, fused up-sampling --
    must stay within twice the distance from that fp32 step that PyTorch-ROCm's own convolution kernels under the same autocast show,
    loss and every parameter gradient.  Smoothed activations on all three runs (tests/_smooth.py: a ReLU kink flipped by an fp16-sized
    perturbation is an O(1) gradient change that says nothing about the kernels)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from types import SimpleNamespace
    from _smooth import smooth_activations
    from openstereo_amd import engine
    from openstereo_amd.models.stereo_models import StereoBase
    from openstereo_amd.utils.weights import synth_images

    def build():
        m = StereoBase(SimpleNamespace(MAX_DISP=192, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128],
                                       N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=False, EVAL_ITERS=32, TRAIN_ITERS=22))
        sd = synth_state_dict(m, seed=41, head_gain=20.0, gain=0.9)
        sd.update({k: v for k, v in synth_state_dict(m, seed=41, head_gain=20.0, gain=0.8).items() if k.startswith("update_block.")})
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        for mod in m.modules():
            if isinstance(mod, nn.modules.batchnorm._BatchNorm):
                mod.eval()
        return m
    H, W = 320, 736
    L, R = synth_images(1, H, W, seed=33, max_shift=40.0)
    L, R = L.to(DEV), R.to(DEV)
    gt = torch.from_numpy(np.random.default_rng(5).uniform(1.0, 120.0, (1, H, W)).astype(np.float32)).to(DEV)

    def step(amp, eager=False):
        m = build()
        scaler = torch.amp.GradScaler("cuda", enabled=amp, init_scale=256.0)
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        with (_stock_torch_convs() if eager else _null()), smooth_activations():
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                out = m({"left": L, "right": R})
                loss, _ = m.get_loss(out, {"disp": gt})
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
        gr = {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
        assert all(torch.isfinite(v).all() for v in gr.values()), "scaled fp16 gradients overflowed"
        return float(loss.detach()), gr, out["disp_pred"].detach().float()

    old = engine.get_precision()
    engine.set_precision("f16x3")
    try:
        l32, g32, d32 = step(False)
        l16, g16, d16 = step(True)
        le, ge, de = step(True, eager=True)
    finally:
        engine.set_precision(old)
    assert g16.keys() == g32.keys() == ge.keys() and len(g32) > 100
    assert abs(l16 - l32) <= 2.0 * abs(le - l32) + 2e-3 * abs(l32), (l32, l16, le)
    epe = lambda a, b: float((a - b).abs().mean())
    assert epe(d16, d32) <= 2.0 * epe(de, d32) + 2e-2, (epe(d16, d32), epe(de, d32))
    worst_eng = worst_ref = 0.0
    bad = {}
    for k in g32:
        s = float(g32[k].abs().max()) + 1e-20
        e_eng, e_ref = float((g16[k] - g32[k]).abs().max()) / s, float((ge[k] - g32[k]).abs().max()) / s
        worst_eng, worst_ref = max(worst_eng, e_eng), max(worst_ref, e_ref)
        if not e_eng <= 2.0 * e_ref + 3e-2:
            bad[k] = (e_eng, e_ref)
    print(f"[whole-model amp step] loss fp32-class {l32:.5f} engine-amp {l16:.5f} eager-amp {le:.5f}; worst relative gradient distance to the fp32-class step: "
          f"engine {worst_eng:.3e}, eager autocast {worst_ref:.3e}; final disparity EPE engine {epe(d16, d32):.2e} eager {epe(de, d32):.2e}")
    assert not bad, bad
    assert worst_eng > 1e-5, "the AMP step must have run the fp16 arithmetic"
