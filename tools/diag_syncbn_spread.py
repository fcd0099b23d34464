"""Where does a FREEZE_BN training step of a convert_sync_batchnorm'ed GwcNet leave the unconverted one?  (tests/test_gpu_syncbn.py)

Runs the step on several model objects in several orders and prints, per pair of runs, the worst parameter-gradient distance; then compares
every leaf module's forward output of one unconverted and one converted instance to name the first module whose output differs.
    python tools/diag_syncbn_spread.py
"""
import copy
import sys

import torch
import torch.nn as nn

sys.path.insert(0, ".")
from openstereo_amd.utils.weights import synth_state_dict, synth_images      # noqa: E402
from openstereo_amd.models.gwcnet import GwcNet                               # noqa: E402

DEV = "cuda"


def freeze(m):
    for x in m.modules():
        if isinstance(x, nn.modules.batchnorm._BatchNorm):
            x.eval()
    return m


def step(net, L, R, taps=None):
    hooks = []
    if taps is not None:
        for name, mod in net.named_modules():
            if not list(mod.children()):
                hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: taps.append((name, type(m).__name__, o.detach().float().clone()))
                                                       if torch.is_tensor(o) else None))
    net.zero_grad(set_to_none=True)
    out = net({"left": L, "right": R})
    loss = sum(p.float().abs().mean() for p in out["disp_preds"])
    loss.backward()
    for h in hooks:
        h.remove()
    return float(loss), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}


def dist(ga, gb):
    worst, key = 0.0, None
    for k in ga:
        s = max(1e-6, float(ga[k].abs().max()))
        e = float((ga[k] - gb[k]).abs().max()) / s
        if e > worst:
            worst, key = e, k
    return worst, key


def poison(value):
    """every torch.empty / empty_like / new_empty issued from Python returns memory filled with `value` (float dtypes on the GPU): an
    engine path that reads what it never wrote shows up as a changed (value = 3e4) or NaN (value = nan) gradient"""
    e0, el0, ne0 = torch.empty, torch.empty_like, torch.Tensor.new_empty

    def fill(t):
        if t.is_cuda and t.is_floating_point() and t.numel():
            t.fill_(value)
        return t
    torch.empty = lambda *a, **k: fill(e0(*a, **k))
    torch.empty_like = lambda *a, **k: fill(el0(*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: fill(ne0(self, *a, **k))


def main_trace(nruns=10):
    """one unconverted model object, the same step `nruns` times: every gradient that passes a leaf module's output (tensor hooks, in the
    order the backward pass produces them) and every parameter gradient; for each run that leaves run 0, the first places it does"""
    m = GwcNet()
    m.load_state_dict(synth_state_dict(m, seed=0))
    L, R = synth_images(1, 64, 128, seed=1)
    L, R = L.to(DEV), R.to(DEV)
    net = freeze(copy.deepcopy(m).train()).to(DEV)
    runs = []
    for it in range(nruns):
        rec, fwd = [], []

        def fhook(mod, inp, out, name=None):
            if torch.is_tensor(out):
                fwd.append((name, out.detach().float().abs().max().item(), out.detach().double().sum().item()))
                if out.requires_grad:
                    out.register_hook(lambda g, name=name: rec.append((name, g.detach().float().clone())))
        hs = [mod.register_forward_hook(lambda a, b, c, n=n: fhook(a, b, c, n)) for n, mod in net.named_modules() if not list(mod.children())]
        net.zero_grad(set_to_none=True)
        out = net({"left": L, "right": R})
        for i, pr in enumerate(out["disp_preds"]):
            pr.register_hook(lambda g, i=i: rec.append((f"pred{i}", g.detach().float().clone())))
        loss = sum(p.float().abs().mean() for p in out["disp_preds"])
        loss.backward()
        for h in hs:
            h.remove()
        runs.append((rec, fwd, {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}))
    base_rec, base_fwd, base_par = runs[0]
    for it in range(1, nruns):
        rec, fwd, par = runs[it]
        fdev = [(n, abs(s - s0) / max(1e-30, abs(s0))) for (n, a, s), (n0, a0, s0) in zip(fwd, base_fwd) if s != s0]
        line = f"run {it}: forward sums differing in {len(fdev)} of {len(fwd)} leaf outputs" + (f" (first {fdev[0][0]} {fdev[0][1]:.1e}, worst {max(e for _, e in fdev):.1e})" if fdev else "")
        assert [n for n, _ in rec] == [n for n, _ in base_rec]
        dev = []
        for idx, ((n, g), (_, g0)) in enumerate(zip(rec, base_rec)):
            e = float((g - g0).abs().max()) / max(1e-30, float(g0.abs().max()))
            if e > 1e-5:
                dev.append((idx, n, e, float(g0.abs().max()), float(g0.abs().median())))
        pdev = [(k, float((par[k] - base_par[k]).abs().max()) / max(1e-30, float(base_par[k].abs().max()))) for k in base_par]
        pdev = [(k, e) for k, e in pdev if e > 1e-5]
        print(line + f"; {len(dev)} of {len(rec)} passing gradients and {len(pdev)} of {len(base_par)} parameter gradients off by > 1e-5")
        for idx, n, e, mx, med in dev[:6]:
            print(f"      backward position {idx}: grad at output of {n}: rel {e:.2e} (max |g| {mx:.2e}, median {med:.2e})")
        for k, e in sorted(pdev, key=lambda t: -t[1])[:4]:
            print(f"      parameter {k}: {e:.2e}")


def main():
    if "det" in sys.argv[2:]:                       # MIOpen's deterministic attribute for the torch convolutions (the 2-D backbone in training mode)
        torch.backends.cudnn.deterministic = True
    if "nobench" in sys.argv[2:]:
        torch.backends.cudnn.benchmark = False
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        return main_trace()
    if len(sys.argv) > 1 and sys.argv[1].startswith("poison="):
        poison(float(sys.argv[1].split("=")[1]))
        return main_poison()
    return main_spread()


def main_poison():
    m = GwcNet()
    m.load_state_dict(synth_state_dict(m, seed=0))
    L, R = synth_images(1, 64, 128, seed=1)
    L, R = L.to(DEV), R.to(DEV)
    nets = [freeze(copy.deepcopy(m).train()).to(DEV) for _ in range(2)]
    runs = [step(nets[i], L, R) for i in (0, 1, 0, 1)]
    print("losses", [r[0] for r in runs])
    names = [k for k, _ in nets[0].named_parameters()]
    for j in (1, 2, 3):
        bad = []
        for k in names:
            if k not in runs[0][1]:
                continue
            a, b = runs[0][1][k], runs[j][1][k]
            s = max(1e-6, float(a.abs().max())) if bool(torch.isfinite(a).all()) else float("nan")
            e = float((a - b).abs().max()) / s
            if not (e <= 1e-5):
                bad.append((k, e))
        print(f"run {j} vs run 0: {len(bad)} of {len(names)} parameter gradients off by > 1e-5; nonfinite in run 0: "
              f"{sum(not bool(torch.isfinite(g).all()) for g in runs[0][1].values())}")
        for k, e in bad[-12:]:
            print(f"    {k}: {e:.3e}")


def main_spread():
    m = GwcNet()
    m.load_state_dict(synth_state_dict(m, seed=0))
    L, R = synth_images(1, 64, 128, seed=1)
    L, R = L.to(DEV), R.to(DEV)
    mk_plain = lambda: freeze(copy.deepcopy(m).train()).to(DEV)
    mk_conv = lambda: nn.SyncBatchNorm.convert_sync_batchnorm(freeze(copy.deepcopy(m).train())).to(DEV)
    nets = {"P1": mk_plain(), "C1": mk_conv(), "P2": mk_plain(), "C2": mk_conv()}
    order = ["P1", "C1", "P2", "C2", "P1", "C1", "C2", "P2"]
    runs = []
    for tag in order:
        loss, g = step(nets[tag], L, R)
        runs.append((tag, loss, g))
        print(f"run {len(runs) - 1} {tag}: loss {loss!r}")
    print("pairwise worst relative gradient distance (run i vs run j):")
    for i in range(len(runs)):
        row = []
        for j in range(len(runs)):
            row.append("%8.1e" % dist(runs[i][2], runs[j][2])[0])
        print(f"  {i} {runs[i][0]}: " + " ".join(row))
    w, k = dist(runs[0][2], runs[1][2])
    print("worst P1 vs C1:", w, k)
    ta, tb = [], []
    step(nets["P1"], L, R, ta)
    step(nets["C1"], L, R, tb)
    print(len(ta), len(tb), "leaf outputs")
    shown = 0
    for (na, ca, a), (nb, cb, b) in zip(ta, tb):
        d = float((a - b).abs().max()) / max(1e-12, float(a.abs().max()))
        if d > 0 and shown < 12:
            print(f"  first differing outputs: {na} ({ca} / {cb}) rel {d:.3e} shape {tuple(a.shape)} stride-class {a.stride()[:2]}")
            shown += 1
    if not shown:
        print("  every leaf-module forward output is bit-identical: the spread is in the backward pass")
    # the same two instances once more, BatchNorm backward isolated: d(input) of one frozen norm fed the same tensor
    name, bn_p = [(n, x) for n, x in nets["P1"].named_modules() if isinstance(x, nn.BatchNorm3d)][0]
    bn_c = dict(nets["C1"].named_modules())[name]
    assert isinstance(bn_c, nn.SyncBatchNorm)
    x = torch.randn(1, bn_p.num_features, 12, 16, 32, device=DEV)
    for fmt, xin in (("contiguous", x), ("channels_last_3d", x.contiguous(memory_format=torch.channels_last_3d))):
        outs = []
        for bn in (bn_p, bn_c):
            xi = xin.clone().requires_grad_()
            y = bn(xi)
            y.square().sum().backward()
            outs.append((y.detach(), xi.grad))
        print(f"  one frozen norm, {fmt}: |dy| {float((outs[0][0] - outs[1][0]).abs().max()):.3e} |dgrad| {float((outs[0][1] - outs[1][1]).abs().max()):.3e}"
              f" (weights equal: {bool(torch.equal(bn_p.weight, bn_c.weight))})")


if __name__ == "__main__":
    main()
