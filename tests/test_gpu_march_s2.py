"""The stride-2 d-marching convolution (csrc/conv_march_s2.h, r6): 3x3x3 stride-2 pad-1 layers with 64 output channels on split tensors --
conv1 of the GwcNet / PSMNet hourglasses (models/gwcnet/hourglass.py:19-24, Conv3d(32, 64, 3, stride 2, pad 1) + BN + ReLU).

Against the layer as the reference computes it (torch fp32 on the CPU: conv + eval BatchNorm + activation), against the brick form of the same
layer (bit 29 of osa_conv_b_ring_mask switches the form), bit-identical when repeated; the launch counter proves which form ran.  Cases cover
ragged H / W (tiles of 4 x 32 output pixels), odd and even D (an even LAST input plane completes an output plane on its own), D cut into
segments and uncut, 2 / 4 input chunks, several batch items, all activations, and the smallest legal volume."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from openstereo_amd.utils.weights import synth_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # name, Ci, (B, D, H, W), activation
    ("hourglass conv1 shape, small", 32, (1, 12, 16, 64), "relu"),
    ("ragged H and W", 32, (2, 8, 11, 37), "relu"),
    ("odd D (even last plane)", 32, (1, 7, 9, 70), "relu"),
    ("odd everything, leaky", 32, (1, 5, 7, 5), "leaky"),
    ("64 input channels (4 chunks), no activation", 64, (1, 6, 10, 66), "none"),
    ("many columns: D cut into segments", 32, (3, 16, 40, 130), "relu"),
    ("D = 2", 32, (1, 2, 8, 64), "relu"),
    ("wide: 5 column tiles", 32, (1, 4, 6, 290), "relu"),
]


def _eye(c):
    m = nn.Conv3d(c, c, 1, bias=False)
    m.weight.data = torch.eye(c).reshape(c, c, 1, 1, 1).clone()
    return m


def _bn(c, name):
    bn = nn.BatchNorm3d(c)
    bn.load_state_dict({k: synth_tensor(f"{name}.{k}", v.shape, 2) for k, v in bn.state_dict().items()})
    return bn.eval()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_stride2_marching_conv_vs_torch_and_brick(case, lib):
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d, is_split
    name, Ci, (B, D, H, W), act = case
    Co = 64
    conv = nn.Conv3d(Ci, Co, 3, 2, 1, bias=False)
    conv.weight.data = synth_tensor(name + ".w", conv.weight.shape, 1)
    bn = _bn(Co, name)
    x = torch.from_numpy(np.random.default_rng(3).normal(0, 1, (B, Ci, D, H, W)).astype(np.float32))
    actf = {"relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.01), "none": lambda t: t}[act]
    with torch.no_grad():
        ref = actf(bn(conv(x)))
        xs = PackedConv3d(_eye(Ci).to(DEV), None, 0, precision="f16x3")(ops.to_cl(x.to(DEV)), out_split=True)
        assert is_split(xs)
        pc = PackedConv3d(conv.to(DEV), bn.to(DEV), {"none": 0, "relu": 1, "leaky": 2}[act], 0.01, precision="f16x3")
        back = PackedConv3d(_eye(Co).to(DEV), None, 0, precision="f16x3")
        n0 = lib.osa_conv3d_march_s2_launches()
        y = pc(xs, out_split=True)
        assert lib.osa_conv3d_march_s2_launches() == n0 + 1, "the layer did not take the stride-2 marching form"
        assert torch.equal(y, pc(xs, out_split=True)), "not deterministic"
        assert is_split(y) and tuple(y.shape[2:]) == tuple(ref.shape[2:])
        got = back(y)[:, :Co].cpu()
        torch.testing.assert_close(got, ref, atol=3e-5, rtol=3e-5, msg=lambda m: f"stride-2 marching conv [{name}] vs torch: {m}")
        # the brick form of the same launch
        mask = lib.osa_conv_b_ring_mask(0)
        lib.osa_conv_b_ring_mask(mask & ~(1 << 29))
        try:
            n1 = lib.osa_conv3d_march_s2_launches()
            yb = pc(xs, out_split=True)
            assert lib.osa_conv3d_march_s2_launches() == n1, "the switch did not select the brick form"
        finally:
            lib.osa_conv_b_ring_mask(mask)
        torch.testing.assert_close(got, back(yb)[:, :Co].cpu(), atol=3e-5, rtol=3e-5, msg=lambda m: f"stride-2 marching conv [{name}] vs brick form: {m}")
        # the range block of the output (max |value| of what was written) must cover the tensor: the next layer scales by it
        from openstereo_amd import ranges
        assert float(ranges.amax_of(ranges.meta_of(y))) >= float(ref.abs().max()) * (1 - 1e-5)


def test_layers_outside_the_form_keep_the_brick_kernel(lib):
    """fp32 (not split) tensors, a residual, 32 or 128 output channels, stride 1: the eligibility test falls through, nothing else changes"""
    from openstereo_amd import ops
    from openstereo_amd.engine import PackedConv3d
    g = torch.Generator().manual_seed(0)
    x = ops.to_cl(torch.randn(1, 32, 6, 8, 40, generator=g).to(DEV))
    with torch.no_grad():
        n0 = lib.osa_conv3d_march_s2_launches()
        PackedConv3d(nn.Conv3d(32, 64, 3, 2, 1, bias=False).to(DEV), None, 1, precision="f16x3")(x)                       # fp32 in / out
        PackedConv3d(nn.Conv3d(32, 128, 3, 2, 1, bias=False).to(DEV), None, 1, precision="f16x3")(x, out_split=True)      # 128 channels
        PackedConv3d(nn.Conv3d(32, 64, 3, 2, 1, bias=False).to(DEV), None, 1, precision="f32")(x)                         # exact-f32 mode
        assert lib.osa_conv3d_march_s2_launches() == n0
