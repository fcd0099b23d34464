"""Host logic of the deferred, batched weight gradients (openstereo_amd/autograd.py, r6) without a GPU: a weight that is applied N times in a
step queues its (x, dy) pairs and gets ONE run() over the queue when the last counted use arrives; uses are grouped by shape key; a use
whose backward never runs is flushed into `.grad` by the end-of-backward callback.  (The kernels behind run() are covered by
tests/test_gpu_channel_sums.py and tests/test_gpu_autograd.py.)"""
import torch

from openstereo_amd import autograd as AG


class _Use(torch.autograd.Function):
    """y = x * w_value (a stand-in for a convolution): backward hands (x, dy) to the deferral machinery exactly as _Conv3d does"""

    @staticmethod
    def forward(ctx, x, w, cache, key, log):
        ctx.save_for_backward(x, w)
        ctx.cache, ctx.key, ctx.log = cache, key, log
        AG._defer_note_use(cache, True, w, None, False, None)
        return x * w.detach().sum()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors

        def run(items):
            ctx.log.append((ctx.key, len(items)))
            return sum((it[0] * it[1]).sum() for it in items) * torch.ones_like(w), None
        dw, _ = AG._defer_wgrad(ctx.cache, ctx.key, (x, dy), run)
        return dy * w.detach().sum(), dw, None, None, None


def _setup():
    w = torch.tensor([0.5, 1.5], requires_grad=True)
    cache = AG._Memo({}, ("stamp",))
    return w, cache, []


def test_three_uses_one_launch_and_the_same_gradient():
    w, cache, log = _setup()
    xs = [torch.full((3,), float(i + 1), requires_grad=True) for i in range(3)]
    loss = sum(_Use.apply(x, w, cache, "shapeA", log).sum() * (i + 1) for i, x in enumerate(xs))
    loss.backward()
    assert log == [("shapeA", 3)]                                   # ONE run over the three queued pairs
    want = sum(float((x.detach() * (i + 1)).sum()) for i, x in enumerate(xs))
    assert torch.allclose(w.grad, torch.full((2,), want))
    st = cache.memo["defer"]
    assert st["uses"] == 0 and st["done"] == 0 and not st["pend"]   # the next step starts from zero
    # second step: same result, again one launch
    w.grad = None
    log.clear()
    loss = sum(_Use.apply(x, w, cache, "shapeA", log).sum() * (i + 1) for i, x in enumerate(xs))
    loss.backward()
    assert log == [("shapeA", 3)] and torch.allclose(w.grad, torch.full((2,), want))


def test_uses_of_different_shapes_are_grouped():
    w, cache, log = _setup()
    xa, xb, xc = (torch.ones(3, requires_grad=True), torch.ones(5, requires_grad=True), torch.ones(3, requires_grad=True))
    loss = _Use.apply(xa, w, cache, "3", log).sum() + _Use.apply(xb, w, cache, "5", log).sum() + _Use.apply(xc, w, cache, "3", log).sum()
    loss.backward()
    assert sorted(log) == [("3", 2), ("5", 1)]
    assert torch.allclose(w.grad, torch.full((2,), 3.0 + 5.0 + 3.0))


def test_a_use_that_is_not_back_propagated_is_flushed_at_the_end_of_the_backward_pass():
    w, cache, log = _setup()
    xa, xb = torch.ones(3, requires_grad=True), torch.ones(4, requires_grad=True)
    ya, yb = _Use.apply(xa, w, cache, "k", log), _Use.apply(xb, w, cache, "k", log)      # two counted uses ...
    ya.sum().backward()                                                                    # ... one backward: the count stays short
    assert log == [("k", 1)]                                        # delivered by the end-of-backward callback, straight into .grad
    assert torch.allclose(w.grad, torch.full((2,), 3.0))
    st = cache.memo["defer"]
    assert st["uses"] == 0 and not st["pend"] and not AG._defer_live
    yb.sum().backward()                                             # the other use, later: a direct (undeferred) gradient, accumulated by autograd
    assert torch.allclose(w.grad, torch.full((2,), 3.0 + 4.0))


def test_single_use_is_not_deferred():
    w, cache, log = _setup()
    x = torch.ones(3, requires_grad=True)
    _Use.apply(x, w, cache, "k", log).sum().backward()
    assert log == [("k", 1)] and torch.allclose(w.grad, torch.full((2,), 3.0))


def test_level_join_delivers_the_accumulated_gradient_once_and_only_for_the_lookups_that_ran():
    """geometry._LevelJoin (r6): an identity placed BEFORE the lookups; the lookups return no level gradient of their own but add into the
    shared accumulator, and autograd's dependency count runs the join after every lookup of the running backward pass -- also a partial one"""
    from openstereo_amd import geometry as G

    class _FakeLookup(torch.autograd.Function):
        @staticmethod
        def forward(ctx, d, state, level):
            ctx.state, ctx.shape = state, level.shape
            ctx.save_for_backward(d)
            return (level * d).sum().reshape(1)

        @staticmethod
        def backward(ctx, dout):
            (d,) = ctx.saved_tensors
            if ctx.state.acc is None:
                ctx.state.acc = [torch.zeros(ctx.shape)]
            ctx.state.acc[0] += dout * d                             # what osa_geo_lookup_bwd_acc_f32 does
            return None, None, None

    for which in ((0, 1, 2), (1,)):
        base = torch.ones(4, requires_grad=True)
        level = base * 2.0                                           # a non-leaf pyramid level, as in CombinedGeoEncodingVolume
        st = G._LevelAcc()
        (joined,) = G._LevelJoin.apply(st, level)
        ds = [torch.full((4,), float(i + 1)) for i in range(3)]
        outs = [_FakeLookup.apply(d, st, joined) for d in ds]
        sum(outs[i] for i in which).sum().backward()
        want = sum(ds[i] for i in which) * 2.0
        assert torch.allclose(base.grad, want) and st.acc is None
