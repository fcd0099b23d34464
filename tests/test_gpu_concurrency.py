"""Kernels that share the GPU with the d-marching convolution (r5).

bench.py's timed configuration runs three sub-batches on three HIP streams, so every kernel of the forward can be co-resident with another
sub-batch's `conv_march_kernel` (250-256 VGPRs, 16-pass f16 MFMAs).  r5 found the fused head returning wrong disparities in isolated quarter
waves (16 pixels of one row) under exactly that co-residency when it was compiled with packed-fp32 math (v_pk_*_f32): its loads were right,
its arithmetic was not (tools/diag_head_under_load.py, profiles/round5/head_packed_math_under_march_load.txt).  softargmin.hip is built
without the SLP vectoriser since (openstereo_amd/build.py EXTRA_FLAGS).  This test is the regression: each VALU kernel of the GwcNet forward
launched repeatedly next to two streams of marching convolutions must return, bit for bit, what it returns on an idle GPU."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
B, D, H, W = 3, 48, 136, 240                      # one sub-batch of the timed configuration at quarter resolution


def _march_load(split):
    """two streams looping the 3x3x3 32 -> 32 layer at 3 pairs in the f16x3 mode: <.., 0, 0> (fp32 tensors) or <.., 1, 1> (split tensors)"""
    from openstereo_amd import engine, ops
    from openstereo_amd.engine import PackedConv3d
    g = torch.Generator().manual_seed(2)
    conv = nn.Conv3d(32, 32, 3, padding=1, bias=False).to(DEV)
    pc0 = PackedConv3d(conv, None, 1, precision="f16x3")
    run = (lambda t: pc0(t, out_split=True)) if split else pc0
    xs = []
    for _ in range(2):
        t = ops.to_cl(torch.randn(B, 32, D, H, W, generator=g).to(DEV))
        t._osa_meta = engine.input_meta(t)
        xs.append(pc0(t, out_split=True) if split else t)
    torch.cuda.synchronize()
    lib = __import__("openstereo_amd._lib", fromlist=["x"]).load()
    n0 = lib.osa_conv3d_march_launches()
    run(xs[0])
    assert lib.osa_conv3d_march_launches() == n0 + 1, "the load must be the d-marching form"
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    return lambda: [run_on(st, run, t) for st, t in zip(streams, xs)]


def run_on(st, run, t):
    with torch.cuda.stream(st):
        for _ in range(3):
            run(t)


@pytest.mark.parametrize("split", [False, True], ids=["march fp32 tensors", "march split tensors"])
@pytest.mark.parametrize("kernel", ["head", "classifier", "volume"])
def test_valu_kernels_next_to_the_marching_conv(kernel, split):
    from openstereo_amd import engine, ops
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        if kernel == "head":
            cost = (torch.randn(B, D, H, W, generator=g) * 3.0).to(DEV)
            launch = lambda: ops.upsample_softargmin(cost, 4 * D, 4 * H, 4 * W)
        elif kernel == "classifier":
            x = ops.to_cl(torch.randn(B, 32, D, H, W, generator=g).to(DEV))
            clf = engine.SmallCoConv3d(nn.Conv3d(32, 1, 3, padding=1, bias=False).to(DEV))
            launch = lambda: clf(x)
        else:
            feat = ops.to_cl(torch.randn(2 * B, 320, 1, H, W, generator=g).to(DEV))
            cat = ops.to_cl(torch.randn(2 * B, 12, 1, H, W, generator=g).to(DEV))
            feat._osa_meta, cat._osa_meta = engine.input_meta(feat), engine.input_meta(cat)
            launch = lambda: ops.build_cost_volume_from_cl(feat, 40, cat, B, D, cat_channels=12, out_split=True)
        ref = launch().clone()                                     # idle GPU
        torch.cuda.synchronize()
        assert torch.equal(launch().view(torch.int32), ref.view(torch.int32))
        load = _march_load(split)
        outs = []
        for _ in range(12):
            load()
            outs.append(launch())
        torch.cuda.synchronize()
        bad = [int((o.view(torch.int32) != ref.view(torch.int32)).sum()) for o in outs]
    assert sum(bad) == 0, f"{kernel}: {sum(bad)} differing 32-bit words in {sum(1 for b in bad if b)} of {len(bad)} launches next to the marching conv"
