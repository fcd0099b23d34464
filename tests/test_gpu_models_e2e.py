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


def test_end_to_end_classes_refuse_cpu():
    """No CPU path: CPU tensors are refused loudly (training mode is covered by tests/test_gpu_autograd.py)."""
    from openstereo_amd.models.stereo_models import LightStereo
    cfg = SimpleNamespace(MAX_DISP=MAXD, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)
    m = LightStereo(cfg).eval()
    L, Rr = _images("cpu")
    with pytest.raises(RuntimeError, match="GPU engine only"):
        m({"left": L, "right": Rr})
