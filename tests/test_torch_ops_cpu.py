"""CPU: `torch.ops.openstereo_amd.*` exist with schemas and meta (fake) kernels -- what tracing / export / torch.compile need --
and have no CPU kernel (no fallback)."""
import pytest
import torch


def test_ops_are_registered_with_schemas_and_fake_kernels():
    import openstereo_amd.torch_ops as T
    for name in T.OPS:
        assert hasattr(torch.ops.openstereo_amd, name), name
    sch = str(torch.ops.openstereo_amd.gwc_volume.default._schema)
    assert "Tensor left" in sch and "num_groups" in sch and sch.endswith("-> Tensor")
    L = torch.empty(2, 320, 136, 240, device="meta")
    v = torch.ops.openstereo_amd.gwc_volume(L, L, 48, 40)
    assert v.shape == (2, 40, 48, 136, 240) and v.device.type == "meta"
    c = torch.ops.openstereo_amd.concat_volume(L[:, :12], L[:, :12], 48, True)
    assert c.shape == (2, 24, 48, 136, 240)
    assert torch.ops.openstereo_amd.corr_volume(L, L, 48).shape == (2, 48, 136, 240)
    cost = torch.empty(2, 1, 48, 136, 240, device="meta", dtype=torch.float16)
    d = torch.ops.openstereo_amd.upsample_softargmin(cost, 192, 544, 960, False)
    assert d.shape == (2, 544, 960) and d.dtype == torch.float16            # fp16 in -> fp16 out
    p = torch.empty(2, 48, 136, 240, device="meta")
    assert torch.ops.openstereo_amd.softargmin(p, True).shape == (2, 1, 136, 240)
    assert torch.ops.openstereo_amd.softmax_softargmin(p, False).shape == (2, 136, 240)
    assert torch.ops.openstereo_amd.context_upsample(p[:, :1], torch.empty(2, 9, 544, 960, device="meta"), 4, True, 4.0).shape == (2, 544, 960)
    with pytest.raises(RuntimeError):
        torch.ops.openstereo_amd.gwc_volume(torch.empty(1, 10, 4, 8, device="meta"), torch.empty(1, 10, 4, 8, device="meta"), 4, 3)


def test_ops_have_no_cpu_kernel():
    import openstereo_amd.torch_ops  # noqa: F401
    x = torch.zeros(1, 8, 4, 8)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.openstereo_amd.gwc_volume(x, x, 4, 2)


def test_ops_trace_through_fake_tensor_mode():
    """An exporter sees the op as one node (deploy/export.py style tracing)."""
    import openstereo_amd.torch_ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        L = torch.empty(1, 32, 16, 32, device="cuda") if torch.cuda.is_available() else torch.empty(1, 32, 16, 32, device="meta")
        v = torch.ops.openstereo_amd.gwc_volume(L, L, 8, 4)
        assert v.shape == (1, 4, 8, 16, 32)
