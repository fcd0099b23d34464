"""CPU: the oracle restatement returns the reference's dtypes under autocast.

tests/test_gpu_autocast.py uses the oracle (run by PyTorch-ROCm under autocast) as "what the unpatched reference composition returns"
on the GPU box, where /root/reference does not exist.  That is legitimate only if the restatement is built from the same torch ops in
the same order as the reference's functions -- autocast's dtype rules are per op.  Here, where the reference is mounted, both run under
CPU autocast (bf16 and fp16) on low-precision and fp32 inputs and must agree in dtype and value."""
import importlib
import os
import sys
import types

import pytest
import torch
import torch.nn.functional as F

REF = os.environ.get("OPENSTEREO_REF", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not mounted")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("low", [True, False])
def test_oracle_functions_follow_the_reference_dtypes_under_autocast(dt, low):
    from openstereo_amd import attach
    from oracle import torch_ref as R
    attach.stub_reference_packages(REF)
    sys.modules.setdefault("timm", types.ModuleType("timm"))
    cv = importlib.import_module("stereo.modeling.cost_volume.cost_volume")
    dr = importlib.import_module("stereo.modeling.disp_pred.disp_regression")
    rf = importlib.import_module("stereo.modeling.disp_refinement.disp_refinement")
    upd = importlib.import_module("stereo.modeling.models.igev.update")
    g = torch.Generator().manual_seed(0)
    c = (lambda t: t.to(dt)) if low else (lambda t: t)
    l, r = c(torch.randn(1, 16, 6, 20, generator=g)), c(torch.randn(1, 16, 6, 20, generator=g))
    with torch.autocast("cpu", dtype=dt), torch.no_grad():
        p = F.softmax(c(torch.randn(1, 8, 6, 20, generator=g)), 1)
        d, w9 = c(torch.rand(1, 1, 6, 20, generator=g) * 10), c(F.softmax(torch.randn(1, 9, 24, 80, generator=g), 1))
        pairs = [("gwc", cv.build_gwc_volume(l, r, 8, 4), R.gwc_volume(l, r, 8, 4)),
                 ("concat", cv.build_concat_volume(l, r, 8), R.concat_volume(l, r, 8)),
                 ("corr", cv.correlation_volume(l, r, 8), R.corr_volume(l, r, 8)),
                 ("corr2", cv.build_corr_volume(l, r, 24), R.build_corr_volume(l, r, 24)),
                 ("regression", dr.disparity_regression(p, 8), R.disparity_regression(p, 8)),
                 ("regression low", dr.disparity_regression(c(p), 8), R.disparity_regression(c(p), 8)),
                 ("context_upsample", rf.context_upsample(d, w9), R.context_upsample(d, w9, 4))]
        # a module-level composition: the IGEV update block (ConvGRU gating, torch.cat promotion in the motion encoder)
        from conftest import igev_update_case
        blk_e, sd, net, inp, corr, disp = igev_update_case()
        args = types.SimpleNamespace(CORR_LEVELS=2, CORR_RADIUS=4, N_GRU_LAYERS=3, N_DOWNSAMPLE=2)
        blk = upd.BasicMultiUpdateBlock(args, hidden_dims=[128, 128, 128]).eval()
        blk.load_state_dict(sd)
        net, inp = [c(t) for t in net], [[c(t) for t in ts] for ts in inp]
        n1, m1, d1 = blk([t.clone() for t in net], inp, corr, disp)
        n2, m2, d2 = R.igev_update_block(net, inp, corr, disp, sd)
        pairs += [(f"net{i}", a, b) for i, (a, b) in enumerate(zip(n1, n2))] + [("mask", m1, m2), ("delta", d1, d2)]
        pairs.append(("motion encoder", blk.encoder(disp, corr), R._motion_encoder(disp, corr, sd, "encoder")))
    for name, a, b in pairs:
        assert a.dtype == b.dtype, (name, a.dtype, b.dtype)
        torch.testing.assert_close(a.float(), b.float(), rtol=0, atol=0, msg=lambda s: f"{name}: {s}")
