import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def lib():
    from openstereo_amd import _lib
    return _lib.load()


def rnd(shape, seed):
    """Same generator as tests/golden/make_golden.py rnd(): inputs of fixtures that store outputs only."""
    import torch
    return torch.from_numpy(np.random.default_rng(seed).normal(0, 1, shape).astype(np.float32))


def lightstereo_case():
    """LightStereo-S aggregation fixture (make_golden.gen_lightstereo): module kwargs, seeded weights, inputs."""
    from openstereo_amd.utils.weights import synth_state_dict
    from openstereo_amd.models.lightstereo import Aggregation
    kw = dict(in_channels=48, left_att=True, blocks=[1, 2, 4], expanse_ratio=4, backbone_channels=[24, 32, 96, 160])
    agg = Aggregation(**kw).eval()
    sd = synth_state_dict(agg, seed=9)
    agg.load_state_dict(sd)
    x = rnd((1, 48, 32, 64), 51)
    feats = [rnd((1, 24, 32, 64), 52), rnd((1, 32, 16, 32), 53), rnd((1, 96, 8, 16), 54), rnd((1, 160, 4, 8), 55)]
    return agg, sd, x, feats


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def igev_update_case():
    """IGEV update-block fixture (make_golden.gen_igev_update): module, seeded weights, inputs."""
    import torch
    from openstereo_amd.utils.weights import synth_state_dict
    from openstereo_amd.models.igev_update import BasicMultiUpdateBlock
    args = _Cfg(CORR_LEVELS=2, CORR_RADIUS=4, N_GRU_LAYERS=3, N_DOWNSAMPLE=2)
    blk = BasicMultiUpdateBlock(args, hidden_dims=[128, 128, 128]).eval()
    sd = synth_state_dict(blk, seed=11)
    blk.load_state_dict(sd)
    H, W = 16, 32
    net = [torch.tanh(rnd((1, 128, H >> i, W >> i), 70 + i)) for i in range(3)]
    inp = [[rnd((1, 128, H >> i, W >> i), 80 + 3 * i + j) * 0.5 for j in range(3)] for i in range(3)]
    corr, disp = rnd((1, 162, H, W), 90), rnd((1, 1, H, W), 91).abs() * 10
    return blk, sd, net, inp, corr, disp


def lightstereo_stage_case():
    """Cost-stage fixture (make_golden.gen_lightstereo, second half): module with the same seeded weights, inputs."""
    import torch
    from openstereo_amd.utils.weights import synth_state_dict
    from openstereo_amd.models.lightstereo import LightStereoCostStage
    agg, sd, _, feats = lightstereo_case()
    st = LightStereoCostStage(max_disp=192).eval()
    st.cost_agg.load_state_dict(sd)
    fl = [rnd((1, 24, 32, 64), 56)] + feats[1:]
    fr0 = torch.roll(fl[0], shifts=-3, dims=3) + 0.1 * rnd((1, 24, 32, 64), 57)
    return st, {"cost_agg." + k: v for k, v in sd.items()}, fl, fr0


def igev_refine_case():
    """Refinement-loop fixture (make_golden.gen_igev_update, second half)."""
    import torch
    from openstereo_amd.utils.weights import synth_state_dict
    from openstereo_amd.models.igev_update import IGEVRefiner
    blk, sd, net, inp, _, _ = igev_update_case()
    args = _Cfg(CORR_LEVELS=2, CORR_RADIUS=4, N_GRU_LAYERS=3, N_DOWNSAMPLE=2, SLOW_FAST_GRU=True)
    ref = IGEVRefiner(args, hidden_dims=[128, 128, 128]).eval()
    ref.update_block.load_state_dict(sd)
    H, W = 16, 32
    ml, mr = rnd((1, 96, H, W), 92), rnd((1, 96, H, W), 93)
    gvol = rnd((1, 8, 12, H, W), 94)
    d0 = rnd((1, 1, H, W), 95).abs() * 3
    return ref, {"update_block." + k: v for k, v in sd.items()}, ml, mr, gvol, net, inp, d0


def igev_at_size_case():
    """BASELINE configs[4] at size (make_golden.gen_at_size): 136x240 quarter resolution, 96-channel matching features, 8 x 48 geometry volume."""
    import torch
    from openstereo_amd.utils.weights import synth_state_dict
    from openstereo_amd.models.igev_update import IGEVRefiner
    args = _Cfg(CORR_LEVELS=2, CORR_RADIUS=4, N_GRU_LAYERS=3, N_DOWNSAMPLE=2, SLOW_FAST_GRU=True)
    ref = IGEVRefiner(args, hidden_dims=[128, 128, 128]).eval()
    sd = synth_state_dict(ref.update_block, seed=11, gain=0.8)      # contractive recurrence, see make_golden.gen_at_size
    ref.update_block.load_state_dict(sd)
    H, W = 136, 240
    ml, mr = rnd((1, 96, H, W), 192), rnd((1, 96, H, W), 193)
    gvol = rnd((1, 8, 48, H, W), 194)
    d0 = rnd((1, 1, H, W), 195).abs() * 3
    net = [torch.tanh(rnd((1, 128, H >> i, W >> i), 170 + i)) for i in range(3)]
    inp = [[rnd((1, 128, H >> i, W >> i), 180 + 3 * i + j) * 0.5 for j in range(3)] for i in range(3)]
    return ref, ml, mr, gvol, net, inp, d0


def stereobase_at_size_case():
    """BASELINE configs[2] at the 320x736 training crop (make_golden.gen_at_size): quarter resolution 80x184, D/4 = 48."""
    from openstereo_amd.utils.weights import synth_state_dict
    from openstereo_amd.models.igev_style import StereoBaseCostStage
    h, w = 80, 184
    st = StereoBaseCostStage(max_disp=192, num_groups=8, concat_channels=8, backbone_channels=[48, 64, 192, 120])
    st.load_state_dict(synth_state_dict(st, seed=8, head_gain=20.0))
    x = (rnd((1, 96, h, w), 201), rnd((1, 96, h, w), 202), rnd((1, 8, h, w), 203), rnd((1, 8, h, w), 204))
    feats = [None, rnd((1, 64, h // 2, w // 2), 205), rnd((1, 192, h // 4, w // 4), 206), rnd((1, 120, h // 8, w // 8), 207)]
    return st.eval(), x, feats
