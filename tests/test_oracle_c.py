"""CPU: the plain-C oracle (oracle/c/oracle.c) against the reference's golden vectors and against the
torch restatement -- two independent statements of the same arithmetic must agree."""
import numpy as np
import torch
import torch.nn.functional as F

from conftest import golden
from oracle import c_ref as Cc
from oracle import torch_ref as O

T = torch.from_numpy


def test_c_volumes_vs_reference_golden():
    g = golden("volumes.npz")
    for tag in ("a", "narrow", "k12"):
        B, C, H, W, D, G = (int(v) for v in g[f"{tag}_meta"])
        np.testing.assert_allclose(Cc.gwc_volume(g[f"{tag}_L"], g[f"{tag}_R"], D, G), g[f"{tag}_gwc"], atol=1e-6)
        np.testing.assert_array_equal(Cc.concat_volume(g[f"{tag}_L"], g[f"{tag}_R"], D), g[f"{tag}_concat"])
        np.testing.assert_array_equal(Cc.concat_volume(g[f"{tag}_L"], g[f"{tag}_R"], D, False), g[f"{tag}_igev_concat"])


def test_c_regression_vs_reference_golden():
    g = golden("regression.npz")
    np.testing.assert_allclose(Cc.softargmin(g["prob"]), g["reg_nokeep"], atol=1e-5)
    np.testing.assert_allclose(Cc.upsample_softargmin(g["low"], 24, 20, 28, False), g["up_false"], atol=2e-5)
    np.testing.assert_allclose(Cc.upsample_softargmin(g["low"], 24, 20, 28, True), g["up_true"], atol=2e-5)
    np.testing.assert_allclose(Cc.upsample_softargmin(g["low2"], 17, 13, 21, False), g["up_odd"], atol=2e-5)


def test_c_conv_family_vs_torch():
    r = np.random.default_rng(0)
    x = r.normal(0, 1, (2, 6, 5, 6, 7)).astype(np.float32)
    for (Co, k, s, p, d) in [(4, 3, 1, 1, 1), (5, 3, 2, 1, 1), (3, 1, 1, 0, 1), (4, 3, 1, 2, 2)]:
        w = r.normal(0, 0.2, (Co, 6, k, k, k)).astype(np.float32)
        ref = F.conv3d(T(x), T(w), None, s, p, d).numpy()
        np.testing.assert_allclose(Cc.conv3d(x, w, s, (p,) * 3, (d,) * 3), ref, atol=2e-5, rtol=1e-5)
    for (Co, k, p, op) in [(4, 3, 1, 1), (3, 4, 1, 0)]:
        w = r.normal(0, 0.2, (6, Co, k, k, k)).astype(np.float32)
        ref = F.conv_transpose3d(T(x), T(w), None, 2, p, op).numpy()
        np.testing.assert_allclose(Cc.deconv3d(x, w, k, 2, p, op), ref, atol=2e-5, rtol=1e-5)
    y = r.normal(0, 1, (2, 4, 3, 4, 5)).astype(np.float32)
    m, v, ga, be = (r.normal(0, 0.1, 4), r.uniform(0.5, 1.5, 4), r.uniform(0.5, 1.5, 4), r.normal(0, 0.1, 4))
    m, v, ga, be = (a.astype(np.float32) for a in (m, v, ga, be))
    res = r.normal(0, 1, y.shape).astype(np.float32)
    ref = F.leaky_relu(F.batch_norm(T(y), T(m), T(v), T(ga), T(be), False, 0.0, 1e-5) + T(res), 0.01).numpy()
    np.testing.assert_allclose(Cc.bn_act(y, m, v, ga, be, 1e-5, res, 2, 0.01), ref, atol=1e-5, rtol=1e-5)


def test_c_hourglass_vs_reference_golden():
    """Compose the gwcnet hourglass (hourglass.py:46-56) from the C primitives."""
    from openstereo_amd.models.gwcnet import Hourglass
    from openstereo_amd.utils.weights import synth_state_dict
    g = golden("gwc_hourglass.npz")
    sd = {k: v.numpy() for k, v in synth_state_dict(Hourglass(8), seed=3).items()}

    def cbn(x, p, stride, pad, act, res=None, deconv=False):
        if deconv:
            y = Cc.deconv3d(x, sd[p + ".0.weight"], 3, 2, 1, 1)
        else:
            k = sd[p + ".0.weight"].shape[2]
            y = Cc.conv3d(x, sd[p + ".0.weight"], stride, (pad,) * 3)
        return Cc.bn_act(y, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"],
                         1e-5, res, act)

    x = g["x"]
    c1 = cbn(x, "conv1.0", 2, 1, 1)
    c2 = cbn(c1, "conv2.0", 1, 1, 1)
    c4 = cbn(cbn(c2, "conv3.0", 2, 1, 1), "conv4.0", 1, 1, 1)
    c5 = cbn(c4, "conv5", 0, 0, 1, res=cbn(c2, "redir2", 1, 0, 0), deconv=True)
    c6 = cbn(c5, "conv6", 0, 0, 1, res=cbn(x, "redir1", 1, 0, 0), deconv=True)
    np.testing.assert_allclose(c6, g["y"], atol=3e-5, rtol=3e-5)
