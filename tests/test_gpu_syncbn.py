"""SYNC_BN models (missing #1 / weak #1).

Every BASELINE config sets `SYNC_BN: true` (cfgs/gwcnet/gwcnet_sceneflow.yaml:37, cfgs/stereobase/stereobase_sceneflow.yaml:49,
cfgs/igev/igev_sceneflow_amp.yaml:39) and the reference's trainer converts the model with `nn.SyncBatchNorm.convert_sync_batchnorm`
before DDP (stereo/modeling/trainer_template.py:79-85); its eval epochs then run the converted model (:281-283).  `nn.SyncBatchNorm` is a
`_BatchNorm` but NOT a `BatchNorm2d / 3d`: a pack site that tested for the latter folded no norm at all.  Here every model family is
converted and must (a) still match the goldens the real reference produced, (b) equal the unconverted model bit for bit (the folded
statistics are the same numbers), in eval mode and in a FREEZE_BN training step (common_utils.py:114-120: BatchNorm modules in eval
mode inside a training model), and (c) a norm the engine cannot fold must raise instead of being dropped."""
import copy
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import golden
from openstereo_amd.utils.weights import synth_state_dict, synth_images

pytestmark = pytest.mark.gpu
DEV = "cuda"
PRECS = ["f32", "f16x3"]


def _with_precision(prec, fn):
    from openstereo_amd import engine
    old = engine.get_precision()
    engine.set_precision(prec)
    try:
        return fn()
    finally:
        engine.set_precision(old)


def _sync(m):
    n_bn = sum(isinstance(x, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)) for x in m.modules())
    m = nn.SyncBatchNorm.convert_sync_batchnorm(m)
    n_sync = sum(isinstance(x, nn.SyncBatchNorm) for x in m.modules())      # (>= n_bn: a BatchNorm registered under two names -- ResidualBlock's norm3 /
    # downsample[1] -- is converted once per name; the copies share its parameters and statistics)
    assert n_sync >= n_bn > 0 and not any(isinstance(x, (nn.BatchNorm2d, nn.BatchNorm3d)) for x in m.modules())
    return m


def _epe(a, b):
    return float((a.float().cpu() - torch.from_numpy(np.asarray(b))).abs().mean())


def _e2e_case(which):
    from openstereo_amd.models import stereo_models as SM
    if which == "stereobase":
        cfg = SimpleNamespace(MAX_DISP=64, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128],
                              N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=False, EVAL_ITERS=4, TRAIN_ITERS=4)
        return SM.StereoBase(cfg), 41, False, "stereobase_disp"
    if which == "igev":
        args = SimpleNamespace(MAX_DISP=64, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2,
                               SLOW_FAST_GRU=True, VALID_ITERS=4, TRAIN_ITERS=4, N_DOWNSAMPLE=2)
        return SM.IGEVStereo(args), 43, True, "igev_disp"
    cfg = SimpleNamespace(MAX_DISP=192, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)
    return SM.LightStereo(cfg), 47, False, "lightstereo_disp"


def _e2e_images(scale255):
    L, R = synth_images(1, 128, 256, seed=31, max_shift=12.0)
    if scale255:
        L, R = (L * 40 + 128).clamp(0, 255), (R * 40 + 128).clamp(0, 255)
    return L.to(DEV), R.to(DEV)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("which", ["stereobase", "igev", "lightstereo"])
def test_converted_e2e_models_match_reference_goldens(which, prec):
    """configs[2] / [3] / [4] classes after convert_sync_batchnorm: the goldens of tests/test_gpu_models_e2e.py (the reference's own
    forward), and bit-for-bit the unconverted model."""
    g = golden("e2e_reference.npz")
    m, seed, s255, key = _e2e_case(which)
    m.load_state_dict(synth_state_dict(m, seed=seed, head_gain=20.0, gain=0.9))
    plain = copy.deepcopy(m).eval().to(DEV)
    conv = _sync(m).eval().to(DEV)
    L, R = _e2e_images(s255)
    with torch.no_grad():
        a = _with_precision(prec, lambda: plain({"left": L, "right": R}))["disp_pred"]
        b = _with_precision(prec, lambda: conv({"left": L, "right": R}))["disp_pred"]
    assert _epe(b, g[key]) < 1e-3, _epe(b, g[key])
    assert torch.equal(a, b), f"converted != unconverted: max diff {(a - b).abs().max().item():.3e}"


@pytest.mark.parametrize("prec", PRECS)
def test_converted_gwcnet_matches_reference_golden(prec):
    """configs[1] (cfgs/gwcnet/gwcnet_sceneflow.yaml:37 SYNC_BN: true) -- small golden of the real reference."""
    from openstereo_amd.models.gwcnet import GwcNet
    g = golden("gwcnet_small.npz")
    net = GwcNet()
    net.load_state_dict(synth_state_dict(net, seed=0))
    plain = copy.deepcopy(net).to(DEV).eval()
    conv = _sync(net).to(DEV).eval()
    L, R = synth_images(1, 64, 128, seed=1)
    with torch.no_grad():
        a = _with_precision(prec, lambda: plain({"left": L.to(DEV), "right": R.to(DEV)}))["disp_pred"]
        b = _with_precision(prec, lambda: conv({"left": L.to(DEV), "right": R.to(DEV)}))["disp_pred"]
    assert np.abs(b.cpu().numpy() - g["disp"]).mean() < 1e-3
    assert torch.equal(a, b)


def test_converted_psmnet_matches_reference_golden():
    """configs[0] at its size, converted."""
    from openstereo_amd.models.psmnet import PSMNet, _Cfg
    g = golden("psmnet_256x512.npz")
    net = PSMNet(_Cfg(MAX_DISP=64))
    net.load_state_dict(synth_state_dict(net, seed=0, head_gain=3.0), strict=False)
    net = _sync(net).to(DEV).eval()
    L, R = synth_images(1, 256, 512, seed=1, max_shift=16.0)
    with torch.no_grad():
        out = net({"left": L.to(DEV), "right": R.to(DEV)})
    for d, k in zip(out["train_preds"], ("disp1", "disp2", "disp3")):
        epe = np.abs(d.cpu().numpy() - g[k]).mean()
        assert epe < 1e-3, (k, epe)


def _freeze_bn(module):
    """what stereo/utils/common_utils.py:114-120 does (class NAME contains 'BatchNorm': SyncBatchNorm included)"""
    n = 0
    for m in module.modules():
        if type(m).__name__.find("BatchNorm") != -1:
            m.eval()
            n += 1
    assert n > 0
    return module


@pytest.mark.parametrize("which", ["stereobase", "gwcnet"])
def test_freeze_bn_training_step_converted_equals_unconverted(which):
    """trainer_template.py:79-85 order: freeze_bn, then convert_sync_batchnorm; one forward + backward in training mode.  Losses and
    every parameter gradient of the converted model equal the unconverted model's (frozen SyncBatchNorm = the same affine map)."""
    if which == "stereobase":
        m, seed, s255, _ = _e2e_case("stereobase")
        m.load_state_dict(synth_state_dict(m, seed=seed, head_gain=20.0, gain=0.9))
        L, R = _e2e_images(s255)
    else:
        from openstereo_amd.models.gwcnet import GwcNet
        m = GwcNet()
        m.load_state_dict(synth_state_dict(m, seed=0))
        L, R = synth_images(1, 64, 128, seed=1)
        L, R = L.to(DEV), R.to(DEV)
    # [MI355X] r5, tools/diag_syncbn_spread.py (profiles/round5/training_step_determinism.txt): PyTorch-ROCm's own convolutions (MIOpen: the
    # strided / dilated 2-D layers the engine leaves to torch in training mode) are not run-to-run reproducible by default -- ulp-level
    # differences in the forward pass, which flip the ReLU mask of single near-zero activations and move individual gradients by 1e-4 .. 1e-2
    # of their tensor's maximum between two runs of the SAME model object.  With MIOpen's deterministic attribute every run of the step --
    # any model object, converted or not -- is bit-identical, engine kernels (forward, data gradient, weight gradient) included.
    det0 = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        plain = _freeze_bn(copy.deepcopy(m).train()).to(DEV)
        plain2 = _freeze_bn(copy.deepcopy(m).train()).to(DEV)          # a second unconverted instance: the spread between two model OBJECTS
        conv = _sync(_freeze_bn(copy.deepcopy(m).train())).to(DEV)
        assert all(not x.training for x in conv.modules() if isinstance(x, nn.SyncBatchNorm))      # convert keeps the frozen (eval) flag
        losses, grads = [], []
        for net in (plain, conv, plain2):      # the third run (another unconverted instance) measures the step's own instance-to-instance spread
            net.zero_grad(set_to_none=True)
            out = net({"left": L, "right": R})
            loss = sum(p.float().abs().mean() for p in out["disp_preds"]) + (out["init_disp"].abs().mean() if "init_disp" in out else 0.0)
            loss.backward()
            losses.append(float(loss.detach()))
            grads.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    finally:
        torch.backends.cudnn.deterministic = det0
    assert np.isfinite(losses[0]) and abs(losses[0] - losses[1]) <= 1e-6 * abs(losses[0]), losses
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 10, (len(grads[0]), len(grads[1]))
    worst, worst_noise = 0.0, 0.0
    for k in grads[0]:
        a, b, a2 = grads[0][k], grads[1][k], grads[2][k]
        s = max(1e-6, float(a.abs().max()))
        noise = float((a - a2).abs().max()) / s          # what is left are float atomics in torch's own backward kernels (interpolate, index ops)
        err = float((a - b).abs().max()) / s
        worst, worst_noise = max(worst, err), max(worst_noise, noise)
        assert err <= TOL_FREEZE[which] + 4.0 * noise, (k, err, noise)     # a dropped or mis-folded norm is O(1)
    print(f"[{which}] converted vs unconverted FREEZE_BN step: worst relative gradient difference {worst:.2e} (two unconverted objects: {worst_noise:.2e})")


# GwcNet: measured 0 (bit-identical) over 8 runs of 4 model objects; the bound leaves room for one ulp of the largest element only
TOL_FREEZE = {"gwcnet": 2e-7, "stereobase": 2e-4}


def test_unfoldable_norm_raises_instead_of_being_dropped():
    """an unknown norm at a pack site must raise, not fall through to "no norm"."""
    from openstereo_amd import _lib
    from openstereo_amd.engine import PackedConv3d, ACT_RELU
    from openstereo_amd.models.igev_style import BasicConv3d, _pack_sb
    blk = BasicConv3d(8, 8, norm_layer=nn.BatchNorm3d, act_layer=nn.LeakyReLU, kernel_size=3, padding=1).to(DEV).eval()
    blk.block[1] = nn.GroupNorm(2, 8).to(DEV)
    with pytest.raises(_lib.EngineError):
        _pack_sb(blk)
    conv = nn.Conv3d(8, 8, 3, padding=1).to(DEV)
    with pytest.raises(_lib.EngineError):
        PackedConv3d(conv, nn.GroupNorm(2, 8).to(DEV), ACT_RELU)
    with pytest.raises(_lib.EngineError):
        PackedConv3d(conv, nn.BatchNorm3d(8).to(DEV).train(), ACT_RELU)             # batch statistics cannot be folded
    with pytest.raises(_lib.EngineError):
        PackedConv3d(conv, nn.BatchNorm3d(8, track_running_stats=False).to(DEV).eval(), ACT_RELU)
    p = PackedConv3d(conv, nn.SyncBatchNorm(8).to(DEV).eval(), ACT_RELU)             # SyncBatchNorm folds
    assert p.scale is not None
