"""Diagnostic (GPU): which convolution module's engine route perturbs the whole-model gradients.  The model is built ONCE; each run enables
the engine route for a subset of modules (others: stock convs) and compares every gradient with the all-stock run."""
import contextlib, os, sys
from types import SimpleNamespace
import numpy as np, torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openstereo_amd.utils.weights import synth_state_dict, synth_images
from openstereo_amd import autograd as AG
from openstereo_amd.models import stereo_models as SM
which = sys.argv[1]
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
L, Rr = synth_images(1, 64, 128, seed=31, max_shift=12.0)
if which == "igev":
    L, Rr = (L * 40 + 128).clamp(0, 255), (Rr * 40 + 128).clamp(0, 255)
L, Rr = L.cuda(), Rr.cuda()
gt = torch.from_numpy(np.random.default_rng(3).uniform(1.0, 30.0, (1, 64, 128)).astype(np.float32)).cuda()
names = {id(mod): n for n, mod in m.named_modules()}
allowed, seen = None, set()
real_eligible = AG._eligible
def eligible(mod, x):
    ok = real_eligible(mod, x)
    if ok:
        seen.add(names.get(id(mod), "?"))
    return ok and (allowed is None or allowed(names.get(id(mod), "?")))
AG._eligible = eligible

def grads():
    m.zero_grad(set_to_none=True)
    out = m({"left": L, "right": Rr})
    loss, _ = m.get_loss(out, {"disp": gt})
    loss.backward()
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}, float(loss.detach())

def report(tag, g, ref, n=4):
    errs = sorted(((float((g[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30)), k) for k in ref), reverse=True)
    print(f"{tag:46s} " + "  ".join(f"{e:.1e} {k}" for e, k in errs[:n]))
    return errs[0][0]

allowed = lambda n: False
ref, l0 = grads()
allowed = None
g, l1 = grads(); print("loss stock", l0, "engine", l1); report("all engine", g, ref, 6)
mods = sorted(seen)
print(len(mods), "eligible modules")
for pre in sorted({n.split(".")[0] for n in mods}):
    allowed = lambda n, pre=pre: n.split(".")[0] == pre
    g, _ = grads(); report("only " + pre, g, ref)
for n0 in mods:
    allowed = lambda n, n0=n0: n == n0
    g, _ = grads()
    e = report("only " + n0, g, ref, 3)
