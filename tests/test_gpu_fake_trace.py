"""An engine model traces under FakeTensors without a kernel running ("tracing / ONNX" callers, deploy/export.py; VERDICT r5 missing #2).

The C++ extension registers Meta kernels for every op of `osa_native` (csrc/torch_ext.cpp: shape inference for the functional ops, one boxed
no-op for the in-place launch ops), so `make_fx(tracing_mode="fake")` walks a GwcNet eval forward -- backbone launches, volume builder, the
aggregation chain in the split format, classifier, fused head -- recording `osa_native` ops and launching nothing.  The recorded graph,
run on real tensors, reproduces the eager forward bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net():
    from openstereo_amd.models.gwcnet import GwcNet
    from openstereo_amd.utils.weights import synth_state_dict, synth_images
    net = GwcNet()
    net.load_state_dict(synth_state_dict(net, seed=0))
    net = net.to(DEV).eval()
    L, R = synth_images(1, 64, 128, seed=1)
    return net, L.to(DEV), R.to(DEV)


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_gwcnet_eval_forward_traces_with_fake_tensors(precision, lib):
    from torch.fx.experimental.proxy_tensor import make_fx
    from openstereo_amd import _ext, engine, ranges
    if _ext.load() is None:
        pytest.skip("the C++ extension is not loaded (OSA_TORCH_EXT=0 / OSA_LIB_PATH)")
    old = engine.get_precision()
    engine.set_precision(precision)
    try:
        net, L, R = _net()
        with torch.no_grad():
            f = lambda a, b: net({"left": a, "right": b})["disp_pred"]
            want = f(L, R)                                         # eager: also builds every packed weight (the trace must not pack)
            torch.cuda.synchronize()
            n_march, n_ring = lib.osa_conv3d_march_launches(), lib.osa_conv_b_ring_launches()
            ranges.reset_arenas()                                  # range arenas are module state: the trace gets its own (fake) ones ...
            gm = make_fx(f, tracing_mode="fake", _allow_non_fake_inputs=True)(L, R)
            ranges.reset_arenas()                                  # ... and leaves none behind
            assert lib.osa_conv3d_march_launches() == n_march and lib.osa_conv_b_ring_launches() == n_ring, "the trace launched engine kernels"
            targets = [str(n.target) for n in gm.graph.nodes if n.op == "call_function"]
            ours = [t for t in targets if t.startswith("osa_native.")]
            assert any("conv_ndhwc" in t for t in ours) and any("cost_volume_cl" in t for t in ours) and any("upsample_softargmin" in t for t in ours), sorted(set(ours))
            assert len(ours) > 60, f"only {len(ours)} engine ops in the trace"
            got = gm(L, R)                                         # the recorded graph on real tensors
            torch.cuda.synchronize()
        assert got.shape == want.shape
        assert torch.equal(got, want), f"traced graph differs from eager: max |diff| {float((got - want).abs().max())}"
    finally:
        engine.set_precision(old)
        ranges.reset_arenas()


def test_meta_kernels_cover_every_op_of_the_extension():
    """every operator of `osa_native` has a Meta kernel (functional ops: shape inference; launch ops: the boxed no-op)"""
    from openstereo_amd import _ext
    if _ext.load() is None:
        pytest.skip("the C++ extension is not loaded")
    names = [n for n in dir(torch.ops.osa_native) if not n.startswith("_")]
    schemas = torch._C._jit_get_all_schemas()
    ours = [s for s in schemas if s.name.startswith("osa_native::")]
    assert len(ours) >= 40
    missing = []
    for s in ours:
        if s.name == "osa_native::abi_version":
            continue
        if not torch._C._dispatch_has_kernel_for_dispatch_key(s.name, "Meta"):
            missing.append(s.name)
    assert not missing, missing
