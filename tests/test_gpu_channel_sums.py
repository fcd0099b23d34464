"""osa_channel_sums (r6, csrc/norm.hip): per-channel sums over the positions of a channels-last tensor -- the bias gradient of the
reference's biased convolutions (stereo/modeling/models/igev/update.py:19-26,38-40: autograd of nn.Conv2d(bias=True)) and the backward of
BatchNorm modules in eval mode (FREEZE_BN, stereo/trainer/trainer_template.py:83-85).
(1) the kernel against torch reductions in fp64 for the layouts / dtypes the training loop hands over; (2) `_FrozenBN` inside
`engine_convs` against torch's own eval-mode batch_norm autograd, fp32 and fp16, 2-D and 3-D; (3) a biased convolution applied several
times in one step (the deferred, batched weight / bias gradients of the update block) against torch autograd of F.conv2d; (4) the
multi-tile weight-gradient kernel against the single-tile kernel on the same operands."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rn(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).normal(0, 1, shape) * scale).astype(np.float32))


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last if t.dim() == 4 else torch.channels_last_3d)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 64, 20, 46), (1, 256, 80, 184), (3, 127, 9, 11), (1, 452, 16, 23), (2, 48, 6, 10, 23), (1, 8, 5, 7), (1, 1024, 4, 9)])
def test_channel_sums_vs_torch(shape, dtype):
    from openstereo_amd import ops
    C = shape[1]
    dy, x = rn(shape, 1).to(DEV).to(dtype), (rn(shape, 2) + 0.5).to(DEV).to(dtype)
    if C % 4:                                   # rows padded to a channel quad, as the engine's tensors are
        pad = [0] * (2 * (len(shape) - 2)) + [0, 4 - C % 4]
        dy, x = _cl(F.pad(dy, pad))[:, :C], _cl(F.pad(x, pad))[:, :C]
    else:
        dy, x = _cl(dy), _cl(x)
    assert ops.cl_rows(dy) is not None and ops.cl_rows(x) is not None
    shift, scale = rn((C,), 3).to(DEV), rn((C,), 4).to(DEV)
    red = [0] + list(range(2, len(shape)))
    bc = [1, C] + [1] * (len(shape) - 2)
    want0 = dy.double().sum(red)
    want1 = (dy.double() * (x.double() - shift.double().view(bc))).sum(red)
    tol = (2e-6 if dtype == torch.float32 else 2e-6) * float(dy.double().abs().sum(red).max() + 1)

    s, dx = ops.channel_sums(dy)
    assert s.shape == (1, C) and dx is None
    assert float((s[0].double() - want0).abs().max()) <= tol
    s, dx = ops.channel_sums(dy, x, shift, scale)
    assert s.shape == (2, C)
    assert float((s[0].double() - want0).abs().max()) <= tol
    assert float((s[1].double() - want1).abs().max()) <= 4 * tol * float(1 + x.abs().max() + shift.abs().max())
    want_dx = (dy.float() * scale.view(bc)).to(dtype)
    assert dx.dtype == dtype and dx.shape == dy.shape
    assert torch.equal(dx, want_dx) or float((dx.float() - want_dx.float()).abs().max()) <= 1e-3 * float(want_dx.float().abs().max())
    # determinism: the two-stage sum has a fixed order
    s2, _ = ops.channel_sums(dy, x, shift, scale)
    assert torch.equal(s, s2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("gy_layout", ["cl", "nchw"])
def test_frozen_bn_backward_vs_torch(dims, dtype, gy_layout):
    from openstereo_amd import autograd as AG
    C = 32
    shape = (2, C, 12, 20) if dims == 2 else (2, C, 4, 12, 20)
    bn = (nn.BatchNorm2d if dims == 2 else nn.BatchNorm3d)(C).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(rn((C,), 1).abs() + 0.5); bn.bias.copy_(rn((C,), 2))
        bn.running_mean.copy_(rn((C,), 3)); bn.running_var.copy_(rn((C,), 4).abs() + 0.3)
    bn.eval()
    x0 = _cl(rn(shape, 5).to(DEV).to(dtype))
    gy = rn(shape, 6).to(DEV).to(dtype)
    gy = _cl(gy) if gy_layout == "cl" else gy.contiguous()         # (a contiguous gradient from a torch op takes the torch-op formulas)

    def run(engine):
        bn.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        if engine:
            with AG.engine_convs():
                y = bn(x)
            assert type(y.grad_fn).__name__ == "_FrozenBNBackward"
        else:
            y = bn(x)
        y.backward(gy)
        return y.detach(), x.grad, bn.weight.grad.clone(), bn.bias.grad.clone()

    y0, dx0, dg0, db0 = run(False)
    y1, dx1, dg1, db1 = run(True)
    assert torch.equal(y0, y1)
    rel = lambda a, b: float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))
    assert rel(dx1, dx0) <= (1e-6 if dtype == torch.float32 else 2e-3)
    assert rel(dg1, dg0) <= (1e-5 if dtype == torch.float32 else 2e-3) and rel(db1, db0) <= (1e-5 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("amp", [False, True])
def test_biased_conv_applied_three_times_gradients(amp):
    """the update block's pattern: one biased 3x3 convolution applied to several inputs in one step -- its weight and bias gradients arrive
    once, from ONE batched launch each, and equal torch autograd's accumulated gradients"""
    from openstereo_amd import autograd as AG, engine
    engine.set_precision("f16x3")
    Ci, Co, H, W = 64, 128, 20, 46
    w = (rn((Co, Ci, 3, 3), 1) * 0.05).to(DEV).requires_grad_()
    b = (rn((Co,), 2) * 0.1).to(DEV).requires_grad_()
    xs = [rn((1, Ci, H, W), 10 + i).to(DEV) for i in range(3)]
    gys = [rn((1, Co, H, W), 20 + i).to(DEV) for i in range(3)]
    conv = nn.Conv2d(Ci, Co, 3, padding=1).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w); conv.bias.copy_(b)

    def run(engine_on):
        conv.zero_grad(set_to_none=True)
        ins = [x.clone().requires_grad_() for x in xs]
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            if engine_on:
                with AG.engine_convs():
                    outs = [conv(x) for x in ins]
            else:
                outs = [conv(x) for x in ins]
            loss = sum((o.float() * g).sum() for o, g in zip(outs, gys))
        loss.backward()
        return conv.weight.grad.clone(), conv.bias.grad.clone(), [x.grad.clone() for x in ins]

    w0, b0, dx0 = run(False)
    w1, b1, dx1 = run(True)
    tol = 3e-3 if amp else 2e-5
    rel = lambda a, b_: float((a.float() - b_.float()).abs().max() / (b_.float().abs().max() + 1e-30))
    assert rel(w1, w0) <= tol and rel(b1, b0) <= tol
    assert all(rel(a, c) <= tol for a, c in zip(dx1, dx0))
    # a second step starts from zero (the queue of the first is empty)
    w2, b2, _ = run(True)
    assert torch.equal(w1, w2) and torch.equal(b1, b2)


@pytest.mark.parametrize("prec", ["f16", "f16x3"])
@pytest.mark.parametrize("case", [("2d 384->256", 384, 256, (1, 3, 3), (3, 1, 20, 46)), ("2d 64->64", 64, 64, (1, 3, 3), (2, 1, 24, 40)),
                                  ("2d 100->72 k1", 100, 72, (1, 1, 1), (2, 1, 17, 33)), ("3d 48->48", 48, 48, (3, 3, 3), (2, 6, 12, 23)),
                                  ("3d 96->160", 96, 160, (3, 3, 3), (1, 4, 9, 18))])
def test_multi_tile_wgrad_vs_single_tile(case, prec):
    """wgrad_mt_kernel (NA x NB tiles per workgroup, brick ranges) vs wgrad_f16x3_kernel (one tile, positions split over the waves): same
    operands and arithmetic, different summation order"""
    from openstereo_amd import _lib, autograd as AG, ops
    from openstereo_amd.ranges import input_meta
    _, Ci, Co, k, (B, D, H, W) = case
    lib = _lib.load()
    x = ops.to_cl(rn((B, Ci, D, H, W), 1).to(DEV))
    dy = ops.to_cl((rn((B, Co, D, H, W), 2) * 1e-2).to(DEV))
    pad = tuple(kk // 2 for kk in k)
    mx, mdy = input_meta(x), input_meta(dy)
    default = lib.osa_conv_b_ring_mask(-1)
    lib.osa_conv_b_ring_mask(default)
    assert (default >> 27) & 1
    got = {}
    try:
        for name, mask in (("mt", default), ("single", default & ~(1 << 27))):
            lib.osa_conv_b_ring_mask(mask)
            dw = torch.empty(Co, Ci, *k, device=DEV)
            AG._wgrad(x, dy, dw, B, D, H, W, Ci, D, H, W, Co, k, 1, pad, (1, 1, 1), 0, prec, mx, mdy)
            got[name] = dw.clone()
    finally:
        lib.osa_conv_b_ring_mask(default)
    want = torch.nn.grad.conv3d_weight(x[:, :Ci].cpu().double().contiguous(), (Co, Ci, *k), dy[:, :Co].cpu().double().contiguous(), padding=pad).float().to(DEV)
    scale = float(want.abs().max())
    assert float((got["mt"] - got["single"]).abs().max()) <= 2e-5 * scale
    assert float((got["mt"] - want).abs().max()) <= (3e-3 if prec == "f16" else 2e-5) * scale


@pytest.mark.parametrize("prec", ["f32", "f16x3", "f16"])
@pytest.mark.parametrize("case", [("2d 128->256", 128, 256, (1, 3, 3), (1, 20, 46)), ("2d 64->36 k1", 64, 36, (1, 1, 1), (1, 17, 33)), ("3d 32->32", 32, 32, (3, 3, 3), (4, 10, 19))])
def test_wgrad_over_a_list_equals_the_concatenation(case, prec):
    """osa_conv3d_wgrad_ws_multi: three equally shaped (x, dy) pairs as a list of tensors vs the same pairs concatenated along the batch --
    the same kernels with per-item base pointers: bit-identical"""
    from openstereo_amd import autograd as AG, ops
    from openstereo_amd.ranges import combine_meta, input_meta
    _, Ci, Co, k, (D, H, W) = case
    xs = [ops.to_cl(rn((1, Ci, D, H, W), 30 + i).to(DEV)) for i in range(3)]
    dys = [ops.to_cl((rn((1, Co, D, H, W), 40 + i) * 1e-2).to(DEV)) for i in range(3)]
    pad = tuple(kk // 2 for kk in k)
    mx, md = combine_meta(*[input_meta(t) for t in xs]), combine_meta(*[input_meta(t) for t in dys])
    a, b = torch.empty(Co, Ci, *k, device=DEV), torch.empty(Co, Ci, *k, device=DEV)
    AG._wgrad(xs, dys, a, 3, D, H, W, Ci, D, H, W, Co, k, 1, pad, (1, 1, 1), 0, prec, mx, md)
    AG._wgrad(AG._cat_batch(xs), AG._cat_batch(dys), b, 3, D, H, W, Ci, D, H, W, Co, k, 1, pad, (1, 1, 1), 0, prec, mx, md)
    assert float(a.abs().max()) > 0 and torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_channel_sums_over_a_list(dtype):
    from openstereo_amd import ops
    C = 128
    ts = [_cl(rn((1, C, 20, 46), 50 + i).to(DEV).to(dtype)) for i in range(5)]
    got = ops.channel_sums_list(ts)
    want = sum(t.double().sum((0, 2, 3)) for t in ts)
    assert got.shape == (C,) and float((got.double() - want).abs().max()) <= 2e-6 * float(sum(t.double().abs().sum((0, 2, 3)) for t in ts).max())
    assert torch.equal(got, ops.channel_sums_list(ts))


@pytest.mark.parametrize("layout", ["nchw fp32", "nhwc12 fp16", "nhwc fp32"])
def test_context_upsample_logits_forward_and_gradients(layout):
    """r6: softmax + x4 gain + convex 3x3 up-sampling as one differentiable op (stereobase_gru.py:196-203 per GRU iteration) against the
    torch composition the reference runs (F.softmax + unfold / nearest / weighted sum, disp_refinement.py:194-204) and its autograd"""
    from openstereo_amd import autograd as AG
    B, h, w, sc = 2, 9, 13, 4
    H, W = h * sc, w * sc
    disp0 = (rn((B, 1, h, w), 1).abs() * 20).to(DEV)
    lg0 = rn((B, 9, H, W), 2).to(DEV)
    gy = rn((B, H, W), 3).to(DEV)
    if layout == "nchw fp32":
        make = lambda t: t.clone()
    elif layout == "nhwc fp32":
        make = lambda t: t.clone().contiguous(memory_format=torch.channels_last)
    else:                                     # channel slice of a 12-channel NHWC fp16 buffer: what the engine's transposed conv returns under autocast
        make = lambda t: torch.cat([t, torch.zeros(B, 3, H, W, device=DEV)], 1).half().contiguous(memory_format=torch.channels_last)[:, :9]

    def ref(d, lg):
        spx = F.softmax(lg.float(), 1)
        u = F.unfold(d * 4.0, 3, 1, 1).reshape(B, -1, h, w)
        u = F.interpolate(u, (H, W), mode="nearest").reshape(B, 9, H, W)
        return (u * spx).sum(1)

    outs = []
    for fn in (ref, lambda d, lg: AG.context_upsample_logits(d, lg, 4, 4.0)):
        d = disp0.clone().requires_grad_()
        lg = make(lg0).detach().requires_grad_()
        out = fn(d, lg)
        out.backward(gy)
        outs.append((out.detach(), d.grad.clone(), lg.grad.float().clone()))
        assert lg.grad.dtype == lg.dtype
    (o0, dd0, dl0), (o1, dd1, dl1) = outs
    half = layout.endswith("fp16")
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    assert rel(o1, o0) <= 2e-6
    assert rel(dd1, dd0) <= 1e-5
    assert rel(dl1, dl0) <= (2e-3 if half else 1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("dims", [2, 3])
def test_training_mode_bn_vs_torch(dims, dtype, monkeypatch):
    """r6: BatchNorm with batch statistics inside engine_convs (_TrainBN: osa_channel_sums + osa_channel_affine, forward and backward)
    against torch's own training-mode batch_norm: output, input / affine gradients, running statistics after two steps
    (gwcnet_disp_processor.py:8-19: convbn_3d in train(); cfgs/gwcnet/gwcnet_sceneflow.yaml trains with batch statistics)."""
    import copy
    from openstereo_amd import autograd as AG
    monkeypatch.setattr(AG, "TRAIN_BN", True)          # (off by default: measured slower than torch's kernels on the GwcNet step)
    C = 32
    shape = (2, C, 12, 20) if dims == 2 else (2, C, 4, 12, 20)
    ref = (nn.BatchNorm2d if dims == 2 else nn.BatchNorm3d)(C).to(DEV)
    with torch.no_grad():
        ref.weight.copy_(rn((C,), 1).abs() + 0.5); ref.bias.copy_(rn((C,), 2))
        ref.running_mean.copy_(rn((C,), 3) * 0.1); ref.running_var.copy_(rn((C,), 4).abs() + 0.3)
    eng = copy.deepcopy(ref)
    ref.train(); eng.train()
    outs = []
    for step in range(2):
        x0 = _cl((rn(shape, 5 + step) * 2.0 + rn((1, C) + (1,) * dims, 9)).to(DEV).to(dtype))
        gy = _cl(rn(shape, 7 + step).to(DEV).to(dtype))
        res = []
        for m, on in ((ref, False), (eng, True)):
            m.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_()
            if on:
                with AG.engine_convs():
                    y = m(x)
                assert type(y.grad_fn).__name__ == "_TrainBNBackward"
            else:
                y = m(x)
            y.backward(gy)
            res.append((y.detach().float(), x.grad.float(), m.weight.grad.clone(), m.bias.grad.clone()))
        outs.append(res)
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    tol = 2e-5 if dtype == torch.float32 else 3e-3
    for (r, e) in outs:
        assert rel(e[0], r[0]) <= tol and rel(e[1], r[1]) <= 5 * tol
        assert rel(e[2], r[2]) <= 5 * tol and rel(e[3], r[3]) <= tol
    assert rel(eng.running_mean, ref.running_mean) <= 1e-5 and rel(eng.running_var, ref.running_var) <= 1e-5
    assert int(eng.num_batches_tracked) == int(ref.num_batches_tracked) == 2


@pytest.mark.parametrize("amp", [False, True])
def test_transposed_conv_applied_three_times_gradients(amp):
    """the k = 4 up-sampling head of the update loop (stereobase_gru.py:114-119 spx_gru: ConvTranspose2d(64, 9, 4, 2, 1), once per GRU
    iteration): its weight gradient is ONE batched class-mode launch over the queued pairs, its bias gradient one channel-sums pass per
    use -- equal to torch autograd's accumulated gradients"""
    from openstereo_amd import autograd as AG, engine
    engine.set_precision("f16x3")
    Ci, Co, H, W = 64, 9, 20, 46
    dc = nn.ConvTranspose2d(Ci, Co, 4, 2, 1).to(DEV)
    with torch.no_grad():
        dc.weight.copy_((rn((Ci, Co, 4, 4), 1) * 0.05).to(DEV)); dc.bias.copy_((rn((Co,), 2) * 0.1).to(DEV))
    xs = [rn((1, Ci, H, W), 10 + i).to(DEV) for i in range(3)]
    gys = [rn((1, Co, 2 * H, 2 * W), 20 + i).to(DEV) for i in range(3)]

    def run(engine_on):
        dc.zero_grad(set_to_none=True)
        ins = [x.clone().requires_grad_() for x in xs]
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            if engine_on:
                with AG.engine_convs():
                    outs = [dc(x) for x in ins]
            else:
                outs = [dc(x) for x in ins]
            loss = sum((o.float() * g).sum() for o, g in zip(outs, gys))
        loss.backward()
        return dc.weight.grad.clone(), dc.bias.grad.clone(), [x.grad.clone() for x in ins]

    w0, b0, dx0 = run(False)
    w1, b1, dx1 = run(True)
    tol = 3e-3 if amp else 2e-5
    rel = lambda a, b_: float((a.float() - b_.float()).abs().max() / (b_.float().abs().max() + 1e-30))
    assert rel(w1, w0) <= tol and rel(b1, b0) <= tol and all(rel(a, c) <= tol for a, c in zip(dx1, dx0))
    w2, b2, _ = run(True)
    assert torch.equal(w1, w2) and torch.equal(b1, b2)


@pytest.mark.parametrize("prec", ["f16", "f16x3"])
@pytest.mark.parametrize("case", [("3d 32->32", 32, 32, (2, 10, 20, 24)), ("3d 24->20 ragged", 24, 20, (1, 5, 9, 13)), ("3d 32->32 one plane", 32, 32, (1, 1, 16, 16))])
def test_three_kd_planes_in_one_workgroup_vs_single_tile(case, prec):
    """wgrad_mt_kernel<.., 1, 1, 3>: 3x3x3 layers with one channel tile -- the three kd planes as three groups of waves of ONE workgroup (P tile
    staged once, Q brick once with a d halo) against the three-workgroup single-tile kernel and torch"""
    from openstereo_amd import _lib, autograd as AG, ops
    from openstereo_amd.ranges import input_meta
    _, Ci, Co, (B, D, H, W) = case
    lib = _lib.load()
    x = ops.to_cl(rn((B, Ci, D, H, W), 1).to(DEV))
    dy = ops.to_cl((rn((B, Co, D, H, W), 2) * 1e-2).to(DEV))
    mx, mdy = input_meta(x), input_meta(dy)
    default = lib.osa_conv_b_ring_mask(-1)
    lib.osa_conv_b_ring_mask(default)
    got = {}
    try:
        for name, mask in (("g3", default), ("single", default & ~(1 << 27))):
            lib.osa_conv_b_ring_mask(mask)
            dw = torch.empty(Co, Ci, 3, 3, 3, device=DEV)
            AG._wgrad(x, dy, dw, B, D, H, W, Ci, D, H, W, Co, (3, 3, 3), 1, (1, 1, 1), (1, 1, 1), 0, prec, mx, mdy)
            got[name] = dw.clone()
    finally:
        lib.osa_conv_b_ring_mask(default)
    want = torch.nn.grad.conv3d_weight(x[:, :Ci].cpu().double().contiguous(), (Co, Ci, 3, 3, 3), dy[:, :Co].cpu().double().contiguous(), padding=1).float().to(DEV)
    scale = float(want.abs().max())
    assert float((got["g3"] - got["single"]).abs().max()) <= 2e-5 * scale
    assert float((got["g3"] - want).abs().max()) <= (3e-3 if prec == "f16" else 2e-5) * scale


def test_new_entry_points_fail_loudly_on_bad_arguments():
    """the C ABI reports what it cannot do (EngineError through _lib.call; no silent fallback): misaligned / mis-strided tensors for the channel
    sums, too many list items for the batched weight gradient, an unsupported up-sampling scale"""
    import ctypes
    from openstereo_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1024, device=DEV)
    out, ws = torch.zeros(64, device=DEV), torch.zeros(1 << 16, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    with pytest.raises(_lib.EngineError, match="alignment|strides"):                 # channel stride 6 is not a multiple of 4
        _lib.call("osa_channel_sums", x.data_ptr(), 0, 6, None, 0, 0, None, None, None, 0, 16, 6, out.data_ptr(), ws.data_ptr(), ws.numel() * 4, st)
    with pytest.raises(_lib.EngineError, match="C <= 1024|bad dims"):
        _lib.call("osa_channel_sums", x.data_ptr(), 0, 2048, None, 0, 0, None, None, None, 0, 4, 2048, out.data_ptr(), ws.data_ptr(), ws.numel() * 4, st)
    assert lib.osa_channel_sums_workspace_bytes(16, 2048) == 0 and lib.osa_channel_sums_workspace_bytes(16, 64) > 0
    ptrs = (ctypes.c_void_p * 25)(*([x.data_ptr()] * 25))
    with pytest.raises(_lib.EngineError, match="1..24"):
        _lib.call("osa_conv3d_wgrad_ws_multi", 0, ptrs, ptrs, 25, out.data_ptr(), 25, 1, 4, 4, 4, 4, 1, 4, 4, 4, 4, 1, 3, 3, 1, 0, 1, 1, 1, 1, 1, 0,
                  None, None, 0, 0, ws.data_ptr(), ws.numel() * 4, st)
    strides = (ctypes.c_longlong * 4)(9 * 12 * 12, 12 * 12, 12, 1)
    with pytest.raises(_lib.EngineError, match="scale"):
        _lib.call("osa_context_upsample_logits_f32", x.data_ptr(), x.data_ptr(), 0, strides, out.data_ptr(), 1, 4, 4, 3, 4.0, st)
