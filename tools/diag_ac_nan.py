"""Diagnostic (GPU): first non-finite tensor in the training forward of an end-to-end class under autocast(fp16)."""
import os, sys
from types import SimpleNamespace
import numpy as np, torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openstereo_amd.utils.weights import synth_state_dict, synth_images
from openstereo_amd.models import stereo_models as SM
from openstereo_amd import autograd as AG
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
names = {id(mod): n for n, mod in m.named_modules()}
first = []
def flat(o):
    if isinstance(o, torch.Tensor): yield o
    elif isinstance(o, dict):
        for v in o.values(): yield from flat(v)
    elif isinstance(o, (list, tuple)):
        for v in o: yield from flat(v)
def hook(mod, inp, out):
    for t in flat(out):
        if t.is_floating_point() and not torch.isfinite(t).all() and len(first) < 6:
            ins = [(str(x.dtype)[6:], float(x.float().abs().max())) for x in flat(inp) if x.is_floating_point()]
            first.append((names[id(mod)], type(mod).__name__, str(t.dtype), ins))
for mod in m.modules():
    mod.register_forward_hook(hook)
# engine Functions: report too
for F_ in (AG._Conv3d, AG._ConvTranspose3d, AG._ConvTranspose2d, AG._GwcVolume, AG._ConcatVolume, AG._SoftmaxSoftArgmin):
    orig = F_.apply
    def wrapped(*a, _o=orig, _n=F_.__name__):
        y = _o(*a)
        if not torch.isfinite(y).all() and len(first) < 6:
            first.append((_n, "Function", str(y.dtype), [(str(x.dtype)[6:], float(x.float().abs().max())) for x in a if isinstance(x, torch.Tensor)]))
        return y
    F_.apply = wrapped
L, Rr = synth_images(1, 64, 128, seed=31, max_shift=12.0)
if which == "igev":
    L, Rr = (L * 40 + 128).clamp(0, 255), (Rr * 40 + 128).clamp(0, 255)
with torch.autocast("cuda", dtype=torch.float16):
    out = m({"left": L.cuda(), "right": Rr.cuda()})
print("disp finite:", bool(torch.isfinite(out["disp_pred"]).all()), "init finite:", bool(torch.isfinite(out["init_disp"]).all()))
for f in first:
    print(f)
