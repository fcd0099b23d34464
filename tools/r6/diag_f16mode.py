"""order-dependent y0 != y2 in tests/test_gpu_f16_mode.py::test_autocast_region_selects_the_f16_mode: which step in between changes the result"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
pre = sys.argv[1] if len(sys.argv) > 1 else "tests/test_feature_pyramid.py"
if pre != "none":
    pytest.main([os.path.join(ROOT, pre), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"])
from conftest import rnd
from openstereo_amd import engine
from openstereo_amd.models.gwcnet import Hourglass
from openstereo_amd.utils.weights import synth_state_dict
engine.set_precision("f16x3")
hg = Hourglass(8).eval()
hg.load_state_dict(synth_state_dict(hg, seed=3))
hg = hg.to("cuda:0")
x = rnd((1, 8, 8, 8, 16), 7).to("cuda:0")
def d(a, b):
    return float((a.float() - b.float()).abs().max()), int((a != b).sum())
with torch.no_grad():
    y0 = hg(x)
    ya = hg(x)
    print("repeat f16x3:", d(y0, ya))
    with torch.autocast("cuda", dtype=torch.float16):
        y1 = hg(x)
    yb = hg(x)
    print("after fp16 autocast run:", d(y0, yb))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y3 = hg(x)
    print("bf16 region result vs f16x3:", d(y0, y3))
    yc = hg(x)
    print("after bf16 autocast run:", d(y0, yc), "max |y0|", float(y0.abs().max()))
    for i in range(3):
        print("again:", d(y0, hg(x)))
    # which run is right, and where
    hg2 = Hourglass(8).eval(); hg2.load_state_dict(synth_state_dict(hg2, seed=3)); hg2 = hg2.to("cuda:0")
    engine.set_precision("f32")
    yr = hg2(x)
    engine.set_precision("f16x3")
    hg3 = Hourglass(8).eval(); hg3.load_state_dict(synth_state_dict(hg3, seed=3)); hg3 = hg3.to("cuda:0")
    z0 = hg3(x); z1 = hg3(x)
    print("fresh module: first vs f32", d(z0, yr), " second vs f32", d(z1, yr), " first vs second", d(z0, z1))
    idx = (z0 != z1).nonzero()
    print(idx[:10].tolist())
    for i in idx[:10].tolist():
        print(i, float(z0[tuple(i)]), float(z1[tuple(i)]), float(yr[tuple(i)]))
