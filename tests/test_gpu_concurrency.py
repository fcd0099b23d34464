"""Kernels that share the GPU with the d-marching convolution (r5; widened r6 to every kernel a sub-batch stream can co-run).

bench.py's timed configuration runs three sub-batches on three HIP streams, so every kernel of the forward can be co-resident with another
sub-batch's `conv_march_kernel` (250-256 VGPRs, 16-pass f16 MFMAs).  r5 found the fused head returning wrong disparities in isolated quarter
waves (16 pixels of one row) under exactly that co-residency when it was compiled with packed-fp32 math (v_pk_*_f32): its loads were right,
its arithmetic was not (profiles/round5/head_packed_math_under_march_load.txt; r6's single-instruction probes: tools/diag_pk_probe.py,
profiles/round6/pk_probe_matrix.txt).  Since r6 the whole library is built without packed-fp32 instructions (openstereo_amd/build.py
NO_PACKED_F32; tests/test_isa_lint_cpu.py checks the object code).  This file is the GPU side of that contract:
  * every kernel family a sub-batch stream launches, 200 launches each next to two streams of marching convolutions, for BOTH marching
    instances (fp32 tensors <.., 0, 0> and split tensors <.., 1, 1>), must return bit for bit what it returns on an idle GPU;
  * whole eval forwards of every model family (GwcNet, StereoBase, IGEV, LightStereo) under the same load, likewise."""
from types import SimpleNamespace

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
B, D, H, W = 3, 48, 136, 240                      # one sub-batch of the timed configuration at quarter resolution
LAUNCHES = 200


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

    def queue(n=2):
        for st, t in zip(streams, xs):
            with torch.cuda.stream(st):
                for _ in range(n):
                    run(t)
    return queue


def _bits(t):
    if isinstance(t, (list, tuple)):
        return torch.cat([_bits(x) for x in t])
    if isinstance(t, dict):
        return torch.cat([_bits(v) for k, v in sorted(t.items()) if torch.is_tensor(v)])
    t = t.detach().contiguous()
    return t.view(torch.int32 if t.element_size() == 4 else torch.int16).flatten().to(torch.int32)


def _check(launch, split, what, launches=LAUNCHES, per_round=2):
    """`launch` next to the marching load `launches` times; mismatching words are counted ON the device, in stream order (no host sync between
    launches, nothing kept alive but one counter)"""
    with torch.no_grad():
        ref = _bits(launch()).clone()                               # idle GPU
        torch.cuda.synchronize()
        assert torch.equal(_bits(launch()), ref), f"{what}: not deterministic on an idle GPU"
        queue = _march_load(split)
        bad = torch.zeros(2, dtype=torch.int64, device=DEV)         # [differing words, launches with a difference]
        for _ in range(launches):
            queue(per_round)
            n = (_bits(launch()) != ref).sum()
            bad[0] += n
            bad[1] += (n > 0).to(torch.int64)
        torch.cuda.synchronize()
    words, hit = (int(v) for v in bad.tolist())
    assert words == 0, f"{what}: {words} differing 32-bit words in {hit} of {launches} launches next to the marching conv"


def _kernel_case(name):
    """(launch closure) for one kernel family of the inference forwards, at the timed configuration's sub-batch shape where that is affordable"""
    from openstereo_amd import engine, ops
    from openstereo_amd.engine import PackedConv3d, DepthwiseConv2d, SmallCoConv3d
    from openstereo_amd.models.lightstereo import nchw_to_cl
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    if name == "head x4":
        cost = r(B, D, H, W) * 3.0
        return lambda: ops.upsample_softargmin(cost, 4 * D, 4 * H, 4 * W)
    if name == "head generic":
        cost = r(B, D, H, W) * 3.0
        return lambda: ops.upsample_softargmin(cost, 2 * D, 2 * H, 2 * W)
    if name == "head align_corners":
        cost = r(B, D // 2, H // 2, W // 2) * 3.0
        return lambda: ops.upsample_softargmin(cost, 2 * D, 2 * H, 2 * W, align_corners=True)
    if name == "softmax_softargmin":
        cost = r(B, D, H, W) * 3.0
        return lambda: ops.softmax_disparity_regression(cost, D)
    if name == "softargmin":
        prob = torch.softmax(r(B, D, H, W), 1)
        return lambda: ops.disparity_regression(prob, D)
    if name == "classifier":
        x = ops.to_cl(r(B, 32, D, H, W))
        clf = SmallCoConv3d(nn.Conv3d(32, 1, 3, padding=1, bias=False).to(DEV))
        return lambda: clf(x)
    if name == "volume cl split":
        feat, cat = ops.to_cl(r(2 * B, 320, 1, H, W)), ops.to_cl(r(2 * B, 12, 1, H, W))
        feat._osa_meta, cat._osa_meta = engine.input_meta(feat), engine.input_meta(cat)
        return lambda: ops.build_cost_volume_from_cl(feat, 40, cat, B, D, cat_channels=12, out_split=True)
    if name == "volume cl fp32":
        fl, fr, cl_, cr_ = r(B, 320, H, W), r(B, 320, H, W), r(B, 12, H, W), r(B, 12, H, W)
        return lambda: ops.build_cost_volume_cl(fl, fr, 40, cl_, cr_, maxdisp=D)
    if name == "gwc volume ncdhw":
        fl, fr = r(B, 320, H, W), r(B, 320, H, W)
        return lambda: ops.build_gwc_volume(fl, fr, D, 40)
    if name == "concat volume":
        fl, fr = r(B, 12, H, W), r(B, 12, H, W)
        return lambda: ops.build_concat_volume(fl, fr, D)
    if name == "corr volume":
        fl, fr = r(B, 96, H, W), r(B, 96, H, W)
        return lambda: ops.correlation_volume(fl, fr, D)
    if name == "to_ndhwc / to_ncdhw":
        x = r(B, 32, D // 2, H, W)
        return lambda: ops.to_ncdhw(ops.to_cl(x))
    if name == "context_upsample":
        disp, wts = r(B, 1, H, W).abs() * 10, r(B, 9, 4 * H, 4 * W)
        return lambda: ops.context_upsample(disp, wts, 4, softmax_weights=True, gain=4.0)
    if name == "dwconv 3x3":
        conv, bn = nn.Conv2d(192, 192, 3, padding=1, groups=192, bias=False).to(DEV), nn.BatchNorm2d(192).to(DEV).eval()
        layer, x = DepthwiseConv2d(conv, bn, 3), nchw_to_cl(r(B, 192, 96, 312))
        return lambda: layer(x)
    if name == "dwconv strip":
        conv = nn.Conv2d(48, 48, (1, 7), padding=(0, 3), groups=48).to(DEV)
        layer, x = DepthwiseConv2d(conv, None, 0), nchw_to_cl(r(B, 48, 96, 312))
        return lambda: layer(x)
    if name == "geo lookup":
        from openstereo_amd.geometry import CombinedGeoEncodingVolume
        h, w = 68, 120
        gev = CombinedGeoEncodingVolume(r(1, 96, h, w), r(1, 96, h, w), r(1, 8, D, h, w), num_levels=2, radius=4)
        disp = r(1, 1, h, w).abs() * 8
        coords = torch.arange(w, device=DEV, dtype=torch.float32).view(1, 1, 1, w).expand(1, 1, h, w).contiguous()
        return lambda: gev(disp, coords)
    if name.startswith("conv2d"):                                  # the brick-kernel instances of the 2-D backbone (D = 1)
        ci, co, k, s = {"conv2d 3->32 s2": (3, 32, 3, 2), "conv2d 32->32": (32, 32, 3, 1), "conv2d 64->64": (64, 64, 3, 1),
                        "conv2d 128->128 d2": (128, 128, 3, 1), "conv2d 1x1 320->128": (320, 128, 1, 1)}[name]
        dil = 2 if "d2" in name else 1
        conv = nn.Conv2d(ci, co, k, stride=s, padding=dil * (k // 2), dilation=dil, bias=False).to(DEV)
        bn = nn.BatchNorm2d(co).to(DEV).eval()
        pc = PackedConv3d(conv, bn, 1, precision="f16x3")
        x = nchw_to_cl(r(2 * B, ci, 136 * (4 if ci == 3 else 2 if ci == 32 else 1), 240 * (4 if ci == 3 else 2 if ci == 32 else 1)))
        x._osa_meta = engine.input_meta(x)
        return lambda: pc(x)
    if name.startswith("conv3d") or name.startswith("deconv3d"):   # brick-kernel 3-D instances: stride-2, 1x1x1, transposed, 64 -> 32 V0
        ci, co, k, s, tr, dd, hh, ww = {"conv3d 32->64 s2": (32, 64, 3, 2, False, D, H, W), "conv3d 64->64": (64, 64, 3, 1, False, D // 2, H // 2, W // 2),
                                        "conv3d 64->128 s2": (64, 128, 3, 2, False, D // 2, H // 2, W // 2), "conv3d 1x1 32->32": (32, 32, 1, 1, False, D, H, W),
                                        "deconv3d 128->64": (128, 64, 3, 2, True, D // 4, H // 4, W // 4), "deconv3d 64->32": (64, 32, 3, 2, True, D // 2, H // 2, W // 2),
                                        "conv3d 64->32 (brick or march)": (64, 32, 3, 1, False, D, H, W)}[name]
        conv = (nn.ConvTranspose3d(ci, co, 3, stride=2, padding=1, output_padding=1, bias=False) if tr
                else nn.Conv3d(ci, co, k, stride=s, padding=k // 2, bias=False)).to(DEV)
        bn = nn.BatchNorm3d(co).to(DEV).eval()
        pc = PackedConv3d(conv, bn, 1, precision="f16x3")
        x = ops.to_cl(r(B, ci, dd, hh, ww))
        x._osa_meta = engine.input_meta(x)
        return lambda: pc(x)
    raise KeyError(name)


KERNELS = ["head x4", "head generic", "head align_corners", "softmax_softargmin", "softargmin", "classifier", "volume cl split", "volume cl fp32",
           "gwc volume ncdhw", "concat volume", "corr volume", "to_ndhwc / to_ncdhw", "context_upsample", "dwconv 3x3", "dwconv strip", "geo lookup",
           "conv2d 3->32 s2", "conv2d 32->32", "conv2d 64->64", "conv2d 128->128 d2", "conv2d 1x1 320->128",
           "conv3d 32->64 s2", "conv3d 64->64", "conv3d 64->128 s2", "conv3d 1x1 32->32", "deconv3d 128->64", "deconv3d 64->32", "conv3d 64->32 (brick or march)"]


@pytest.mark.parametrize("split", [False, True], ids=["march fp32 tensors", "march split tensors"])
@pytest.mark.parametrize("kernel", KERNELS)
def test_kernels_next_to_the_marching_conv(kernel, split):
    _check(_kernel_case(kernel), split, kernel)


def _model_case(name):
    from openstereo_amd.utils.weights import synth_state_dict, synth_images
    from openstereo_amd.models import stereo_models as SM
    from openstereo_amd.models.gwcnet import GwcNet
    h, w, maxd = 128, 256, 64
    base = dict(MAX_DISP=maxd, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2)
    if name == "GwcNet":
        m, scale = GwcNet(), False
    elif name == "StereoBase":
        m, scale = SM.StereoBase(SimpleNamespace(NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, SLOW_FAST_GRU=False, EVAL_ITERS=4, **base)), False
    elif name == "IGEV":
        m, scale = SM.IGEVStereo(SimpleNamespace(SLOW_FAST_GRU=True, VALID_ITERS=4, N_DOWNSAMPLE=2, **base)), True
    else:
        m, scale = SM.LightStereo(SimpleNamespace(MAX_DISP=maxd, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)), False
    m.load_state_dict(synth_state_dict(m, seed=41, head_gain=20.0, gain=0.9) if name != "GwcNet" else synth_state_dict(m, seed=0))
    m = m.to(DEV).eval()
    L, R = synth_images(2, h, w, seed=31, max_shift=12.0)
    if scale:
        L, R = (L * 40 + 128).clamp(0, 255), (R * 40 + 128).clamp(0, 255)
    L, R = L.to(DEV), R.to(DEV)
    return lambda: m({"left": L, "right": R})["disp_pred"]


@pytest.mark.parametrize("split", [False, True], ids=["march fp32 tensors", "march split tensors"])
@pytest.mark.parametrize("model", ["GwcNet", "StereoBase", "IGEV", "LightStereo"])
def test_whole_forwards_next_to_the_marching_conv(model, split):
    """every kernel of an eval forward of each model family (2 pairs at 128x256: backbone instances, layout copies, volume, 3-D / 2-D aggregation,
    GRU loop with lookups, convex upsampling, heads) while two streams loop the marching convolution: 60 forwards, bit-identical to the idle result.
    The stand-in 2-D trunks of StereoBase / IGEV / LightStereo are torch (MIOpen) modules: MIOpen's convolutions differ by ulps run to run on an
    idle GPU unless its deterministic attribute is set (r5: profiles/round5/training_step_determinism.txt), so the test sets it."""
    det0 = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        _check(_model_case(model), split, f"{model} forward", launches=60, per_round=3)
    finally:
        torch.backends.cudnn.deterministic = det0
