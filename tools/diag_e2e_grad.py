"""Diagnostic (GPU): end-to-end training gradients of the engine classes vs the reference-autograd golden, with parts of the engine
swapped for stock PyTorch-ROCm ops to localise a discrepancy.   python tools/diag_e2e_grad.py igev|stereobase"""
import contextlib
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openstereo_amd.utils.weights import synth_state_dict, synth_images          # noqa: E402
from openstereo_amd import autograd as AG, geometry as GEO                      # noqa: E402
from openstereo_amd.models import stereo_models as SM                            # noqa: E402
from oracle import torch_ref as R                                                 # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "igev"
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "e2e_reference_train.npz"))


def build():
    if which == "stereobase":
        m, seed = SM.StereoBase(SimpleNamespace(MAX_DISP=64, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128],
                                                N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=False, EVAL_ITERS=4, TRAIN_ITERS=3)), 41
    else:
        m, seed = SM.IGEVStereo(SimpleNamespace(MAX_DISP=64, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2,
                                                SLOW_FAST_GRU=True, VALID_ITERS=4, TRAIN_ITERS=3, N_DOWNSAMPLE=2)), 43
    m.load_state_dict(synth_state_dict(m, seed=seed, head_gain=20.0, gain=0.9))
    m = m.cuda().train()
    for mod in m.modules():
        if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm3d)):
            mod.eval()
    return m


def run(tag):
    m = build()
    L, Rr = synth_images(1, 64, 128, seed=31, max_shift=12.0)
    if which == "igev":
        L, Rr = (L * 40 + 128).clamp(0, 255), (Rr * 40 + 128).clamp(0, 255)
    gt = torch.from_numpy(np.random.default_rng(3).uniform(1.0, 30.0, (1, 64, 128)).astype(np.float32)).cuda()
    out = m({"left": L.cuda(), "right": Rr.cuda()})
    loss, _ = m.get_loss(out, {"disp": gt})
    loss.backward()
    params = dict(m.named_parameters())
    errs = {}
    for k in g.files:
        if k.startswith(f"{which}_grad::"):
            name = k.split("::", 1)[1]
            want = torch.from_numpy(g[k])
            got = params[name].grad.detach().reshape(-1)[:want.numel()].cpu()
            errs[name] = float((got - want).abs().max() / (want.abs().max() + 1e-20))
    print(f"--- {tag}: loss {float(loss):.6f} (ref {float(g[which + '_loss']):.6f})")
    for k, v in errs.items():
        print(f"   {v:9.2e}  {k}")


run("engine (as shipped)")

# (a) stock convolutions: engine_convs() becomes a no-op
real_ec = AG.engine_convs
AG.engine_convs = contextlib.nullcontext
import openstereo_amd.models.igev_update as UP, openstereo_amd.models.igev_style as IS  # noqa: E402
run("torch convs inside engine_convs() regions (conv_module / volume / lookup still engine)")
AG.engine_convs = real_ec


# (b) torch lookup: the oracle's differentiable composition instead of _Lookup
class TorchGeo:
    def __init__(self, f1, f2, gv, num_levels=2, radius=4):
        self.o = R.GeoEncodingVolume(f1.float(), f2.float(), gv.float(), num_levels=num_levels, radius=radius)
        self.meta = None

    def __call__(self, disp, coords):
        with torch.device("cuda"):
            return self.o(disp, coords)


real_geo = GEO.CombinedGeoEncodingVolume
GEO.CombinedGeoEncodingVolume = TorchGeo
run("torch lookup (oracle composition on the GPU), engine convs")
AG.engine_convs = contextlib.nullcontext
run("torch lookup + torch convs in engine_convs() regions")
