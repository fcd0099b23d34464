"""The PyTorch-ROCm C++ extension over the C ABI (csrc/torch_ext.cpp, openstereo_amd/_ext.py): builds in-tree, registers
`torch.ops.osa_native.*`, has no CPU backend, and on the GPU gives bit-identical results to the ctypes path it replaces."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_extension_builds_loads_and_has_no_cpu_backend(lib):
    from openstereo_amd import _ext, _lib
    path = _ext.build()
    assert os.path.exists(path)
    ns = _ext.load()
    assert ns is not None and int(ns.abi_version()) == lib.osa_abi_version() == _lib.abi_version()
    for name in ("gwc_volume", "concat_volume", "corr_volume", "softargmin", "softmax_softargmin", "upsample_softargmin", "context_upsample", "conv_ndhwc"):
        assert hasattr(ns, name), name
    x = torch.zeros(1, 8, 4, 8)
    with pytest.raises((NotImplementedError, RuntimeError)):        # no CPU kernel is registered: the dispatcher refuses
        ns.gwc_volume(x, x, 4, 2)
    # the shared object links the in-tree C-ABI library through $ORIGIN (it travels with the tree, nothing installed)
    dyn = subprocess.check_output(["readelf", "-d", path], text=True)
    assert "libopenstereo_amd.so" in dyn and "$ORIGIN" in dyn


@pytest.mark.gpu
def test_extension_and_ctypes_paths_agree_bit_for_bit():
    """Same kernels, two dispatch layers: a subprocess with OSA_TORCH_EXT=0 (ctypes) and this process (extension) run GwcNet 64x128 in both
    arithmetic modes plus the functional ops; the results must be identical."""
    import numpy as np
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from openstereo_amd import _ext, engine, ops
from openstereo_amd.models.gwcnet import GwcNet
from openstereo_amd.utils.weights import synth_state_dict, synth_images
out = {}
L, R = synth_images(1, 64, 128, seed=1)
for prec in ("f32", "f16x3"):
    engine.set_precision(prec)
    net = GwcNet(); net.load_state_dict(synth_state_dict(net, seed=0)); net = net.cuda().eval()
    with torch.no_grad():
        out["disp_" + prec] = net({"left": L.cuda(), "right": R.cuda()})["disp_pred"].cpu().numpy()
g = torch.Generator().manual_seed(3)
a, b = torch.randn(2, 16, 6, 20, generator=g).cuda(), torch.randn(2, 16, 6, 20, generator=g).cuda()
out["corr"] = ops.correlation_volume(a, b, 8).cpu().numpy()
c = torch.randn(2, 8, 6, 20, generator=g).cuda()
out["sm"] = ops.softmax_disparity_regression(c, 8).cpu().numpy()
out["up"] = ops.upsample_softargmin(c, 32, 24, 80).cpu().numpy()
out["ext"] = np.array([_ext.load() is not None])
np.savez(sys.argv[1], **out)
''' % ROOT
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, env in (("ext", {}), ("ctypes", {"OSA_TORCH_EXT": "0"})):
            f = os.path.join(td, tag + ".npz")
            subprocess.check_call([sys.executable, "-c", code, f], env={**os.environ, **env})
            res[tag] = dict(np.load(f))
    assert bool(res["ext"]["ext"][0]) and not bool(res["ctypes"]["ext"][0])
    for k in res["ext"]:
        if k != "ext":
            assert np.array_equal(res["ext"][k], res["ctypes"][k]), k
