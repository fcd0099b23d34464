"""which layer of Hourglass(8) differs between the FIRST f16x3 forward after other tests and the later ones"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
pytest.main([os.path.join(ROOT, "tests/test_feature_pyramid.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"])
from conftest import rnd
from openstereo_amd import engine, ops
from openstereo_amd.models import gwcnet as G
from openstereo_amd.utils.weights import synth_state_dict
engine.set_precision("f16x3")
hg = G.Hourglass(8).eval()
hg.load_state_dict(synth_state_dict(hg, seed=3))
hg = hg.to("cuda:0")
x = rnd((1, 8, 8, 8, 16), 7).to("cuda:0")
def stages(xc):
    p = hg._pack()
    out = {}
    c1 = p["c1"](xc); out["c1"] = c1.clone()
    c2 = p["c2"](c1); out["c2"] = c2.clone()
    c3 = p["c3"](c2); out["c3"] = c3.clone()
    c4 = p["c4"](c3); out["c4"] = c4.clone()
    fuse = G._FUSE_REDIR and p["c5"].precision != "f16"
    print("fuse", fuse, "r2.Ci", p["r2"].Ci, "r1.Ci", p["r1"].Ci)
    if fuse and p["r2"].Ci <= 64:
        c5 = p["c5"](c4, redir=(p["r2"], c2))
    else:
        c5 = p["c5"](c4, residual=p["r2"](c2))
    out["c5"] = c5.clone()
    if fuse and p["r1"].Ci <= 32:
        c6 = p["c6"](c5, redir=(p["r1"], xc))
    else:
        c6 = p["c6"](c5, residual=p["r1"](xc))
    out["c6"] = c6.clone()
    return out
with torch.no_grad():
    y0 = hg(x); y1 = hg(x)
    ne = (y0 != y1)
    print("full forward first vs second:", int(ne.sum()))
    for i in ne.nonzero()[:8].tolist():
        print("   ", i, float(y0[tuple(i)]), float(y1[tuple(i)]))
    hg2 = G.Hourglass(8).eval(); hg2.load_state_dict(synth_state_dict(hg2, seed=3)); hg2 = hg2.to("cuda:0")
    xc = ops.to_cl(x)
    z0 = hg2.forward_cl(xc); z0c = z0.clone(); n0 = ops.to_ncdhw(z0, channels=8)
    z1 = hg2.forward_cl(xc); z1c = z1.clone(); n1 = ops.to_ncdhw(z1, channels=8)
    print("second module, cl outputs differ:", int((z0c != z1c).sum()), " ncdhw differ:", int((n0 != n1).sum()), " vs first module's later result:", int((n1 != y1).sum()), int((n0 != y1).sum()))
