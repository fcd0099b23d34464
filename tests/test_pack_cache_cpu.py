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
        engine.set_precision("f16x3" if old == "f32" else "f32")               # arithmetic mode is part of the key
        assert get() == 7
    finally:
        engine.set_precision(old)
    assert get() == 8 and get() == 8
    m._packed = None                                                           # reset_engine() protocol still works
    assert get() == 9
