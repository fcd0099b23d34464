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
