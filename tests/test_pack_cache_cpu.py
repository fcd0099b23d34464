"""CPU: engine.cached_pack rebuilds a module's packed form exactly when a source tensor changed."""
import torch
import torch.nn as nn


def test_cached_pack_invalidation_rules():
    from openstereo_amd import engine
    m = nn.Sequential(nn.Conv3d(4, 4, 3, bias=False), nn.BatchNorm3d(4)).eval()
    builds = []
    get = lambda: engine.cached_pack(m, "_packed", lambda: builds.append(1) or len(builds))
    assert get() == 1 and get() == 1 and len(builds) == 1                     # cached
    with torch.no_grad():
        m[0].weight.mul_(2.0)                                                  # optimiser-style in-place update
    assert get() == 2
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})       # checkpoint load (in-place copy_)
    assert get() == 3
    m.train(); m(torch.randn(2, 4, 5, 5, 5)); m.eval()                         # BN running statistics moved
    assert get() == 4
    m[0].weight = nn.Parameter(torch.zeros_like(m[0].weight))                  # Parameter object replaced
    assert get() == 5
    m.double(); m.float()                                                      # _apply: new storage
    assert get() == 6
    old = engine.get_precision()
    try:
        engine.set_precision("f16x3" if old == "f32" else "f32")               # one slot per arithmetic mode
        assert get() == 7 and get() == 7
    finally:
        engine.set_precision(old)
    assert get() == 6 and len(builds) == 7                                     # flipping back neither rebuilds nor frees the other mode's packs
    with torch.no_grad():
        m[0].weight.add_(1.0)                                                  # a parameter update invalidates EVERY mode's slot
    assert get() == 8
    try:
        engine.set_precision("f16x3" if old == "f32" else "f32")
        assert get() == 9
    finally:
        engine.set_precision(old)
    m._packed = None                                                           # reset_engine() protocol still works
    assert get() == 10


def test_autograd_weight_memo_follows_the_parameter_not_its_fp32_copy(monkeypatch):
    """autograd._pack used to key its memo on the `_version` of the fp32-contiguous WORKING COPY, which is a fresh
    tensor (version 0 for ever) whenever the Parameter is channels_last / fp16 / bf16 -> stale packed weights after an optimizer step.
    The stamp now comes from the Parameter itself: (_version, data_ptr, device, dtype)."""
    from openstereo_amd import autograd as AG
    from openstereo_amd.ops import _f32c
    calls = []
    monkeypatch.setattr(AG, "_pack_now", lambda w, Ci, Co, k, mode, prec: calls.append((mode, float(w.detach().sum()))) or (w.clone(), 1.0))
    for make in (lambda: nn.Parameter(torch.randn(8, 4, 3, 3).to(memory_format=torch.channels_last)),
                 lambda: nn.Parameter(torch.randn(8, 4, 3, 3).bfloat16()),
                 lambda: nn.Parameter(torch.randn(8, 4, 3, 3))):
        w = make()
        calls.clear()
        run = lambda mode="fwd": AG._pack(_f32c(w), 4, 8, (3, 3), mode, "f32", AG._wcache(w))
        run(); run()
        assert len(calls) == 1                                                  # memoised between steps
        run("dgrad_s1"); run("dgrad_s1")
        assert len(calls) == 2                                                  # one pack per role
        with torch.no_grad():
            w.add_(1.0)                                                         # optimizer step
        got, _ = run()
        assert len(calls) == 3 and torch.equal(got, _f32c(w))                   # repacked from the NEW values
        run()
        assert len(calls) == 3
        w.data = w.data.clone()                                                 # storage moved (.to(), load with assign): data_ptr is in the stamp
        run()
        assert len(calls) == 4


def test_engine_convs_is_thread_local():
    """inside `engine_convs()` the conv forwards are rerouted for the entering thread only; another thread (validation,
    DataParallel replica) keeps the stock forward, and concurrent enter / exit restores the class attributes exactly once."""
    import threading
    from openstereo_amd import autograd as AG
    orig = nn.Conv2d.forward
    seen = {}
    inside, release = threading.Event(), threading.Event()

    def other():
        inside.wait(10)
        seen["active_in_other_thread"] = AG.engine_convs.active()
        seen["patched_class_attr"] = nn.Conv2d.forward is not orig
        conv = nn.Conv2d(4, 4, 3, padding=1)
        seen["out"] = conv(torch.zeros(1, 4, 5, 5)).shape               # CPU tensor, stock forward: must simply work
        with AG.engine_convs():                                           # nested use from a second thread
            seen["active_nested"] = AG.engine_convs.active()
        release.set()
    t = threading.Thread(target=other)
    t.start()
    with AG.engine_convs():
        assert AG.engine_convs.active()
        inside.set()
        assert release.wait(10)
        assert nn.Conv2d.forward is not orig                             # still patched: this thread is still inside
    t.join()
    assert nn.Conv2d.forward is orig and not AG.engine_convs.active()
    assert seen == {"active_in_other_thread": False, "patched_class_attr": True, "out": torch.Size([1, 4, 5, 5]), "active_nested": True}


def test_engine_convs_leaves_layers_without_a_backward_kernel_on_torch():
    """`_eligible` checks the weight-gradient / data-gradient preconditions up front (wgrad LDS brick, even dims for the
    stride-2 dgrad), so a dilated conv that the wgrad kernel cannot take stays a torch op instead of raising inside backward()."""
    from openstereo_amd import autograd as AG
    x = torch.zeros(1, 8, 1, 1, requires_grad=True)
    fake = lambda t: type("T", (), {"shape": t, "requires_grad": True})()
    ok = nn.Conv2d(8, 8, 3, padding=1)
    assert AG._wgrad_ok(ok, fake((1, 8, 64, 64)))
    wide = nn.Conv2d(8, 8, 3, padding=12, dilation=12)                    # 1x16x16 brick + 24-pixel halo: > 160 KB of LDS
    assert not AG._wgrad_ok(wide, fake((1, 8, 64, 64)))
    s2 = nn.Conv3d(8, 8, 3, stride=2, padding=1)
    assert AG._wgrad_ok(s2, fake((1, 8, 8, 16, 16))) and not AG._wgrad_ok(s2, fake((1, 8, 7, 16, 16)))
    with torch.no_grad():
        assert AG._wgrad_ok(wide, fake((1, 8, 64, 64)))                   # no gradient can be asked for: nothing to check
