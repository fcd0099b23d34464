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

store, order, fwd_store = {}, [], {}
def _hook_out(name):
    calls = [0]
    def fwd(mod, inp, out):
        k = (name, calls[0]); calls[0] += 1
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for i, o in enumerate(outs):
            if isinstance(o, torch.Tensor) and o.requires_grad:
                kk = k + (i,)
                order.append(kk)
                fwd_store[kk] = o.detach().clone()
                o.register_hook(lambda g, kk=kk: store.__setitem__(kk, g.detach().clone()))
    return fwd, calls
counters = []
for n, mod in m.named_modules():
    if n and (n.startswith("update_block") or n in ("conv", "desc", "cnet", "spx_2_gru", "spx_gru", "cost_agg", "classifier")):
        f, c = _hook_out(n); mod.register_forward_hook(f); counters.append(c)

POISON = None
def grads():
    m.zero_grad(set_to_none=True)
    if POISON is not None:
        junk = [torch.full((n,), POISON, device="cuda") for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12)]
        del junk
    store.clear(); order.clear()
    for c in counters: c[0] = 0
    out = m({"left": L, "right": Rr})
    loss, _ = m.get_loss(out, {"disp": gt})
    loss.backward()
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}, float(loss.detach())

def report(tag, g, ref, n=4):
    errs = sorted(((float((g[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30)), k) for k in ref), reverse=True)
    print(f"{tag:46s} " + "  ".join(f"{e:.1e} {k}" for e, k in errs[:n]))
    return errs[0][0]


def snap():
    g, l = grads()
    return dict(fwd_store), dict(store), list(order), g
for route, al in (("stock", (lambda n: False)), ("engine", None)):
    allowed = al
    POISON = None; A = snap(); B = snap()
    POISON = float("nan"); C = snap()
    POISON = 3e4; D = snap()
    for tag, X in (("repeat", B), ("NaN-poisoned allocator", C), ("3e4-poisoned allocator", D)):
        fw = [kk for kk in A[2] if not torch.equal(A[0][kk], X[0][kk])]
        bw = [kk for kk in reversed(A[2]) if kk in A[1] and kk in X[1] and not torch.equal(A[1][kk], X[1][kk])]
        gp = [k for k in A[3] if not torch.equal(A[3][k], X[3][k])]
        print(f"{route:7s} run vs {tag:26s}: forward outputs differing {len(fw)} (first in forward order: {fw[:3]}); output grads differing {len(bw)} "
              f"(first in backward order: {bw[:3]}); parameter grads differing {len(gp)} of {len(A[3])}")
        if fw:
            kk = fw[0]; print("      first differing forward output max abs diff", float((A[0][kk] - X[0][kk]).abs().max()), "of", float(A[0][kk].abs().max()))
