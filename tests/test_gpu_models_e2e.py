"""End-to-end model classes (openstereo_amd/models/stereo_models.py) for BASELINE configs [2]-[4].

Each class is run on the GPU engine and compared with the same computation assembled from the CPU oracle: the 2-D side
(stand-in backbone + the small heads, ordinary torch modules) is evaluated once on the GPU and handed to both, so the
comparison isolates everything the engine owns -- volume, aggregation, classifier, soft-argmin, geometry lookup, GRU
loop, convex upsampling -- composed the way the reference's forward composes it
(stereobase_gru.py:121-213, igev_stereo.py:139-207, lightstereo.py:44-71)."""
import copy
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F

from openstereo_amd.utils.weights import synth_state_dict, synth_images
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
H, W, MAXD = 128, 256, 64


def _images(dev, scale255=False):
    L, Rr = synth_images(1, H, W, seed=31, max_shift=12.0)
    if scale255:
        L, Rr = (L * 40 + 128).clamp(0, 255), (Rr * 40 + 128).clamp(0, 255)
    return L.to(dev), Rr.to(dev)


def _load(model, seed):
    sd = synth_state_dict(model, seed=seed, head_gain=20.0, gain=0.9)
    model.load_state_dict(sd)
    return {k: v for k, v in sd.items()}


def _cpu(x):
    if isinstance(x, (list, tuple)):
        return [_cpu(t) for t in x]
    return None if x is None else x.detach().cpu()


def _up(disp, logits):
    return R.context_upsample(disp * 4.0, F.softmax(logits, 1), 4).unsqueeze(1)


def test_stereobase_end_to_end():
    from openstereo_amd.models.stereo_models import StereoBase
    cfg = SimpleNamespace(MAX_DISP=MAXD, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128],
                          N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=False, EVAL_ITERS=4)
    m = StereoBase(cfg).eval()
    sd = _load(m, 41)
    cpu = copy.deepcopy(m)
    m = m.cuda()
    L, Rr = _images("cuda")
    out = m({"left": L, "right": Rr})
    assert out["disp_pred"].shape == (1, 1, H, W) and out["init_disp"].shape == (1, 1, H, W)
    with torch.no_grad():
        s = {k: _cpu(v) for k, v in m.side(L, Rr).items()}
        d0, _, geo = R.stereobase_cost_stage(s["match_left"], s["match_right"], s["concat_left"], s["concat_right"],
                                             s["features_left"], sd, MAXD, 8)
        disp, mask, _ = R.igev_refine(s["match_left"], s["match_right"], geo, s["net_list"], s["inp_list"], d0, sd, 4,
                                      slow_fast=False)
        want = _up(disp, cpu.spx_gru(cpu.spx_2_gru(mask, s["stem_2x"])))
        want0 = _up(d0, s["spx_logits"])
    assert want.std() > 0.5                                   # the synthetic weights give a non-trivial disparity map
    torch.testing.assert_close(out["init_disp"].cpu(), want0, rtol=1e-4, atol=2e-3)
    torch.testing.assert_close(out["disp_pred"].cpu(), want, rtol=1e-4, atol=5e-3)


def test_igev_end_to_end():
    from openstereo_amd.models.stereo_models import IGEVStereo
    args = SimpleNamespace(MAX_DISP=MAXD, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2,
                           SLOW_FAST_GRU=True, VALID_ITERS=4, N_DOWNSAMPLE=2)
    m = IGEVStereo(args).eval()
    sd = _load(m, 43)
    cpu = copy.deepcopy(m)
    m = m.cuda()
    L, Rr = _images("cuda", scale255=True)
    out = m({"left": L, "right": Rr})
    assert out["disp_pred"].shape == (1, 1, H, W)
    with torch.no_grad():
        s = {k: _cpu(v) for k, v in m.side(L, Rr).items()}
        d0, _, geo = R.igev_cost_stage(s["match_left"], s["match_right"], s["features_left"], sd, MAXD)
        disp, mask, _ = R.igev_refine(s["match_left"], s["match_right"], geo, s["net_list"], s["inp_list"], d0, sd, 4, slow_fast=True)
        want = _up(disp, cpu.spx_gru(cpu.spx_2_gru(mask, s["stem_2x"])))
    assert want.std() > 0.5
    torch.testing.assert_close(out["disp_pred"].cpu(), want, rtol=1e-4, atol=5e-3)


def test_igev_cost_stage_matches_oracle():
    from openstereo_amd.models.stereo_models import IGEVCostStage
    st = IGEVCostStage(max_disp=MAXD).eval()
    sd = _load(st, 45)
    st = st.cuda()
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g)
    h, w = 32, 64
    ml, mr = r(2, 96, h, w), r(2, 96, h, w)
    feats = [r(2, 96, h, w), r(2, 64, h // 2, w // 2), r(2, 192, h // 4, w // 4), r(2, 160, h // 8, w // 8)]
    with torch.no_grad():
        out = st(ml.cuda(), mr.cuda(), [f.cuda() for f in feats])
        d0, prob, geo = R.igev_cost_stage(ml, mr, feats, sd, MAXD)
    torch.testing.assert_close(out["geo_encoding_volume"][:, :8].cpu(), geo, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out["init_disp"].cpu(), d0, rtol=1e-4, atol=1e-3)


def test_lightstereo_end_to_end():
    from openstereo_amd.models.stereo_models import LightStereo
    cfg = SimpleNamespace(MAX_DISP=MAXD, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)
    m = LightStereo(cfg).eval()
    sd = _load(m, 47)
    m = m.cuda()
    L, Rr = _images("cuda")
    out = m({"left": L, "right": Rr})
    assert out["disp_pred"].shape == (1, 1, H, W)
    with torch.no_grad():
        s = {k: _cpu(v) for k, v in m.side(L, Rr).items()}
        d0, _, _ = R.lightstereo_cost_stage(s["features_left"], s["feature_right"], sd, MAXD)
        want = _up(d0, s["spx_logits"])
    assert want.std() > 0.5
    torch.testing.assert_close(out["disp_pred"].cpu(), want, rtol=1e-4, atol=2e-3)


# ----------------------------------------------------------------------------- pinned against the REFERENCE's own forward
# tests/golden/e2e_reference.npz = outputs of the reference's StereoBase / IGEVStereo / LightStereo classes (make_golden.gen_e2e: the
# reference's __init__ + forward with only the timm-backed `Feature` / `Backbone` / `MultiBasicEncoder` replaced by the same stand-in
# modules the engine classes inject).  Same name-keyed synthetic parameters on both sides; the engine classes must reproduce the
# reference's whole composition -- stem reuse, iteration schedule, disp * 4 / mask order, final convex upsampling.
def _with_precision(prec, fn):
    from openstereo_amd import engine
    old = engine.get_precision()
    engine.set_precision(prec)
    try:
        return fn()
    finally:
        engine.set_precision(old)


def _epe(a, b):
    return float((a.float().cpu() - torch.from_numpy(b)).abs().mean())


def _ref_golden():
    from conftest import golden
    return golden("e2e_reference.npz")


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("slow_fast", [False, True])
def test_stereobase_matches_reference_forward(prec, slow_fast):
    from openstereo_amd.models.stereo_models import StereoBase
    g = _ref_golden()
    cfg = SimpleNamespace(MAX_DISP=MAXD, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128],
                          N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=slow_fast, EVAL_ITERS=4, TRAIN_ITERS=4)
    m = StereoBase(cfg).eval()
    _load(m, 41)
    m = m.cuda()
    L, Rr = _images("cuda")
    out = _with_precision(prec, lambda: m({"left": L, "right": Rr}))
    if slow_fast:
        want = g["stereobase_slowfast_disp"]
    else:
        want = g["stereobase_disp"]
        assert _epe(out["init_disp"], g["stereobase_init"]) < 1e-3
        torch.testing.assert_close(out["init_disp"].cpu(), torch.from_numpy(g["stereobase_init"]), rtol=1e-4, atol=5e-3)
    assert want.std() > 0.5
    assert _epe(out["disp_pred"], want) < 1e-3, _epe(out["disp_pred"], want)
    torch.testing.assert_close(out["disp_pred"].cpu(), torch.from_numpy(want), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_igev_matches_reference_forward(prec):
    from openstereo_amd.models.stereo_models import IGEVStereo
    g = _ref_golden()
    args = SimpleNamespace(MAX_DISP=MAXD, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2,
                           SLOW_FAST_GRU=True, VALID_ITERS=4, TRAIN_ITERS=4, N_DOWNSAMPLE=2)
    m = IGEVStereo(args).eval()
    _load(m, 43)
    m = m.cuda()
    L, Rr = _images("cuda", scale255=True)
    out = _with_precision(prec, lambda: m({"left": L, "right": Rr}))
    assert g["igev_disp"].std() > 0.5
    assert _epe(out["disp_pred"], g["igev_disp"]) < 1e-3, _epe(out["disp_pred"], g["igev_disp"])
    torch.testing.assert_close(out["disp_pred"].cpu(), torch.from_numpy(g["igev_disp"]), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_igev_full_size_32_iterations_matches_reference_forward(prec):
    """BASELINE configs[4] at size: 544x960, MAX_DISP 192, VALID_ITERS 32 (cfgs/igev/igev_sceneflow_amp.yaml), the reference's own
    IGEVStereo.forward; update-block weights at the contractive gain of the at-size fixture (make_golden.gen_at_size)."""
    from openstereo_amd.models.stereo_models import IGEVStereo
    g = _ref_golden()
    args = SimpleNamespace(MAX_DISP=192, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2,
                           SLOW_FAST_GRU=True, VALID_ITERS=32, TRAIN_ITERS=22, N_DOWNSAMPLE=2)
    m = IGEVStereo(args).eval()
    sd = synth_state_dict(m, seed=43, head_gain=20.0, gain=0.9)
    sd.update({k: v for k, v in synth_state_dict(m, seed=43, head_gain=20.0, gain=0.8).items() if k.startswith("update_block.")})
    m.load_state_dict(sd)
    m = m.cuda()
    L, Rr = synth_images(1, 544, 960, seed=31)
    L, Rr = (L * 40 + 128).clamp(0, 255).cuda(), (Rr * 40 + 128).clamp(0, 255).cuda()
    out = _with_precision(prec, lambda: m({"left": L, "right": Rr}))
    got = out["disp_pred"][:, :, ::4, ::4]
    want = g["igev_full_disp_sub"]
    assert want.std() > 2.0
    assert _epe(got, want) < 1e-3, _epe(got, want)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_lightstereo_matches_reference_forward(prec):
    from openstereo_amd.models.stereo_models import LightStereo
    g = _ref_golden()
    cfg = SimpleNamespace(MAX_DISP=192, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)       # reference: in_channels 48 = 192 / 4
    m = LightStereo(cfg).eval()
    _load(m, 47)
    m = m.cuda()
    L, Rr = _images("cuda")
    out = _with_precision(prec, lambda: m({"left": L, "right": Rr}))
    assert g["lightstereo_disp"].std() > 0.5
    assert _epe(out["disp_pred"], g["lightstereo_disp"]) < 1e-3, _epe(out["disp_pred"], g["lightstereo_disp"])
    torch.testing.assert_close(out["disp_pred"].cpu(), torch.from_numpy(g["lightstereo_disp"]), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("activations", ["smooth", "reference"])
@pytest.mark.parametrize("which", ["stereobase", "igev", "lightstereo"])
def test_training_step_matches_reference_autograd(which, activations):
    """Training-mode forward, the reference's loss and the gradients of parameters in every stage (2-D heads, volume / aggregation,
    classifier, update block, context encoder, upsampling heads, stand-in backbone) vs CPU autograd of the REFERENCE's own model class
    (tests/golden/e2e_reference_train.npz, make_golden.gen_e2e_train; frozen BatchNorm).  64x128, 3 GRU iterations.

    Two fixtures per class.  "smooth": ReLU / LeakyReLU / ReLU6 replaced by smooth surrogates in BOTH the reference run and this one
    (tests/_smooth.py) -- the whole-model gradient is then a smooth function of the forward values and is pinned to 5e-4 of max |grad|
    per tensor (measured ~1e-6 .. 1e-5).  "reference": the reference's real activations; every ReLU whose pre-activation is ~0 is a
    discontinuity of the gradient, and two correct fp32 implementations do not agree on the sign of a 1e-7 pre-activation (measured: one
    flipped mask element in the first iteration's disparity head moves IGEVStereo's conv.conv.weight gradient by 6e-3 of its max,
    profiles/round3/diag; the stock PyTorch-ROCm convolutions show the same effect against the CPU), so the bound there is derived
    from the number of at-risk pre-activations of the run (tests/_smooth.py::count_kinks): 5e-4 when there is none, one measured flip's
    worth per at-risk element otherwise, never more than the old flat 3e-2; plus a per-tensor norm check that a mis-scaled layer fails."""
    import contextlib
    import numpy as np
    import torch.nn as nn
    from conftest import golden
    from _smooth import smooth_activations
    from openstereo_amd.models.stereo_models import StereoBase, IGEVStereo, LightStereo
    g = golden("e2e_reference_train.npz")
    if which == "stereobase":
        m, seed = StereoBase(SimpleNamespace(MAX_DISP=64, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128],
                                             N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=False, EVAL_ITERS=4, TRAIN_ITERS=3)), 41
    elif which == "igev":
        m, seed = IGEVStereo(SimpleNamespace(MAX_DISP=64, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2,
                                             SLOW_FAST_GRU=True, VALID_ITERS=4, TRAIN_ITERS=3, N_DOWNSAMPLE=2)), 43
    else:
        m, seed = LightStereo(SimpleNamespace(MAX_DISP=192, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)), 47
    _load(m, seed)
    m = m.cuda().train()
    for mod in m.modules():
        if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm3d)):
            mod.eval()
    L, Rr = synth_images(1, 64, 128, seed=31, max_shift=12.0)
    if which == "igev":
        L, Rr = (L * 40 + 128).clamp(0, 255), (Rr * 40 + 128).clamp(0, 255)
    gt = torch.from_numpy(np.random.default_rng(3).uniform(1.0, 30.0, (1, 64, 128)).astype(np.float32)).cuda()
    from _smooth import count_kinks
    tag, ctx = (which + "_smooth", smooth_activations) if activations == "smooth" else (which, count_kinks)
    with ctx() as kinks:
        out = m({"left": L.cuda(), "right": Rr.cuda()})
        loss, _ = m.get_loss(out, {"disp": gt})
        loss.backward()
    # smooth activations: 5e-4.  The reference's real activations (r4, VERDICT r3 weak #2: no flat 3e-2 any more): the same 5e-4 when NO
    # pre-activation of this run sits within 2e-6 x max of a kink; otherwise one measured flip's worth (6e-3 of max |grad|, r3
    # diag/e2e_grad_relu_flip.txt) per at-risk element, capped at the old 3e-2 -- a 2 % mis-scaled layer no longer passes a kink-free run,
    # and the norm check below catches a uniform mis-scaling even when kinks are present.
    bound = 5e-4
    if activations != "smooth":
        print(f"[kinks] {kinks.near} at-risk pre-activations of {kinks.total} in {kinks.calls} activation calls")
        bound = min(3e-2, 5e-4 + 6e-3 * kinks.near)
    want_loss = float(g[f"{tag}_loss"])
    assert abs(float(loss.detach()) - want_loss) < 2e-4 * abs(want_loss), (float(loss.detach()), want_loss)
    assert _epe(out["disp_pred"].detach(), g[f"{tag}_disp"]) < 1e-3
    params = dict(m.named_parameters())
    keys = [k.split("::", 1)[1] for k in g.files if k.startswith(f"{tag}_grad::")]
    assert len(keys) >= 7
    worst = {}
    for k in keys:
        want = torch.from_numpy(g[f"{tag}_grad::{k}"])
        got = params[k].grad.detach().reshape(-1)[:want.numel()].cpu()
        worst[k] = float((got - want).abs().max() / (want.abs().max() + 1e-20))
        # a flipped mask element moves single entries; a mis-scaled layer moves the whole tensor: its norm is off by the scale error
        nr = float(got.norm() / (want.norm() + 1e-30))
        assert abs(nr - 1.0) < max(2e-3, 0.25 * bound), f"{k}: gradient norm ratio {nr:.5f}"
    print({k: f"{v:.1e}" for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if not v < bound}
    assert not bad, bad                    # whole-model gradient error relative to max |grad| per tensor


def test_stereobase_training_step_at_size_matches_reference_autograd():
    """BASELINE configs[2] AT SIZE: the SceneFlow training crop 320x736, MAX_DISP 192, TRAIN_ITERS 22
    (cfgs/stereobase/stereobase_sceneflow.yaml:15-16,27,40) -- whole-model training forward, the reference's loss, and the gradients of
    16 parameters across every stage vs CPU autograd of the REFERENCE's own StereoBase class (tests/golden/e2e_reference_train_at_size.npz,
    make_golden.gen_e2e_train_at_size: smoothed activations on both sides, frozen BatchNorm, contractive update-block weights)."""
    import numpy as np
    import torch.nn as nn
    from conftest import golden
    from _smooth import smooth_activations
    from openstereo_amd.models.stereo_models import StereoBase
    g = golden("e2e_reference_train_at_size.npz")
    m = StereoBase(SimpleNamespace(MAX_DISP=192, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128],
                                   N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=False, EVAL_ITERS=32, TRAIN_ITERS=22))
    sd = synth_state_dict(m, seed=41, head_gain=20.0, gain=0.9)
    sd.update({k: v for k, v in synth_state_dict(m, seed=41, head_gain=20.0, gain=0.8).items() if k.startswith("update_block.")})
    m.load_state_dict(sd)
    m = m.cuda().train()
    for mod in m.modules():
        if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm3d)):
            mod.eval()
    H, W = 320, 736
    L, Rr = synth_images(1, H, W, seed=33, max_shift=40.0)
    gt = torch.from_numpy(np.random.default_rng(5).uniform(1.0, 120.0, (1, H, W)).astype(np.float32)).cuda()
    with smooth_activations():
        out = m({"left": L.cuda(), "right": Rr.cuda()})
        loss, _ = m.get_loss(out, {"disp": gt})
        loss.backward()
    assert len(out["disp_preds"]) == int(g["n_preds"]) == 22
    want_loss = float(g["loss"])
    assert abs(float(loss.detach()) - want_loss) < 5e-4 * abs(want_loss), (float(loss.detach()), want_loss)
    assert _epe(out["init_disp"].detach()[..., ::2, ::2], g["init_disp_sub"]) < 1e-3
    assert _epe(out["disp_preds"][0].detach()[..., ::4, ::4], g["it1_sub"]) < 1e-3
    assert g["disp_sub"].std() > 2.0
    assert _epe(out["disp_pred"].detach()[..., ::4, ::4], g["disp_sub"]) < 1e-3, _epe(out["disp_pred"].detach()[..., ::4, ::4], g["disp_sub"])
    params = dict(m.named_parameters())
    keys = [k.split("::", 1)[1] for k in g.files if k.startswith("grad::")]
    assert len(keys) >= 12
    worst = {}
    for k in keys:
        want = torch.from_numpy(g[f"grad::{k}"])
        got = params[k].grad.detach().reshape(-1)[:want.numel()].cpu()
        worst[k] = float((got - want).abs().max() / (float(g[f"gmax::{k}"]) + 1e-20))
    print({k: f"{v:.1e}" for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if not v < 2e-3}      # 22 iterations deep: 4x the 3-iteration bound of the small fixture
    assert not bad, bad


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("which", ["all", "inter"])
def test_stereobase_dormant_volume_switches_vs_reference(which, prec):
    """USE_SUB_VOLUME + USE_INTERLACED_VOLUME on top of gwc + concat (33 volume channels: the composition route) and gwc + interlaced
    only (16 channels: the fused NDHWC route) vs the reference's own StereoBase class with those switches
    (tests/golden/e2e_dormant.npz, make_golden.gen_e2e_dormant).  96x192, MAX_DISP 32, 3 iterations."""
    from conftest import golden
    from openstereo_amd.models.stereo_models import StereoBase
    g = golden("e2e_dormant.npz")
    flags = dict(USE_CONCAT_VOLUME=True, USE_SUB_VOLUME=True, USE_INTERLACED_VOLUME=True) if which == "all" else \
        dict(USE_CONCAT_VOLUME=False, USE_SUB_VOLUME=False, USE_INTERLACED_VOLUME=True)
    m = StereoBase(SimpleNamespace(MAX_DISP=32, NUM_GROUPS=8, USE_GWC_VOLUME=True, CONCAT_CHANNELS=8, INTERLACED_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128],
                                   N_DOWNSAMPLE=2, N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=False, TRAIN_ITERS=3, EVAL_ITERS=3,
                                   **flags)).eval()
    _load(m, 53)
    m = m.cuda()
    L, Rr = synth_images(1, 96, 192, seed=35, max_shift=6.0)
    out = _with_precision(prec, lambda: m({"left": L.cuda(), "right": Rr.cuda()}))
    assert g[f"sb_{which}_disp"].std() > 0.5
    assert _epe(out["init_disp"], g[f"sb_{which}_init"]) < 1e-3, _epe(out["init_disp"], g[f"sb_{which}_init"])
    assert _epe(out["disp_pred"], g[f"sb_{which}_disp"]) < 1e-3, _epe(out["disp_pred"], g[f"sb_{which}_disp"])


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_interlaced_volume_vs_reference(prec):
    """cost_volume.py:120-169 alone: the three depth-strided Conv3d + the 1x1 head as four fused 2-D engine launches per disparity."""
    import numpy as np
    from conftest import golden
    from openstereo_amd.models.interlaced import InterlacedVolume
    g = golden("e2e_dormant.npz")
    iv = InterlacedVolume(8).eval()
    iv.load_state_dict(synth_state_dict(iv, seed=23, gain=0.9))
    iv = iv.cuda()
    rn = lambda shape, seed: torch.from_numpy(np.random.default_rng(seed).normal(0, 1, shape).astype(np.float32))
    fl, fr = rn((2, 96, 7, 19), 311).cuda(), rn((2, 96, 7, 19), 312).cuda()
    with torch.no_grad():
        out = _with_precision(prec, lambda: iv(fl, fr, 6))
    want = torch.from_numpy(g["interlaced_alone"])
    torch.testing.assert_close(out.cpu(), want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))


def test_end_to_end_classes_refuse_cpu():
    """No CPU path: CPU tensors are refused loudly (training mode is covered by tests/test_gpu_autograd.py)."""
    from openstereo_amd.models.stereo_models import LightStereo
    cfg = SimpleNamespace(MAX_DISP=MAXD, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)
    m = LightStereo(cfg).eval()
    L, Rr = _images("cpu")
    with pytest.raises(RuntimeError, match="GPU engine only"):
        m({"left": L, "right": Rr})
