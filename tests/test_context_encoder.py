"""MultiBasicEncoder (openstereo_amd/models/context_encoder.py) vs the reference's own class (tests/golden/context_encoder.npz,
make_golden.gen_context_encoder: models/igev/extractor.py:194-297 == models/stereobase/gru_blocks.py:62-148, norm_fn='batch', downsample=2)."""
import os

import pytest
import torch

from conftest import golden
from openstereo_amd.utils.weights import synth_state_dict, synth_images

HD = [128, 128, 128]


def _enc():
    from openstereo_amd.models.context_encoder import MultiBasicEncoder
    enc = MultiBasicEncoder(output_dim=[HD, HD], norm_fn="batch", downsample=2).eval()
    enc.load_state_dict(synth_state_dict(enc, seed=19, gain=0.9))
    return enc


def test_torch_composition_equals_the_reference_class_on_cpu():
    """The mirror's plain-torch forward (training / fallback path) is the reference's: bit-identical outputs on CPU, and with the reference
    mounted also identical state_dict keys."""
    g = golden("context_encoder.npz")
    enc = _enc()
    img, _ = synth_images(2, 64, 128, seed=33, max_shift=8.0)
    with torch.no_grad():
        o = enc._forward_torch(img, False, 3)
        od = enc._forward_torch(img, True, 3)
    for j, lvl in enumerate(("04", "08", "16")):
        for i in range(2):
            assert torch.equal(o[j][i], torch.from_numpy(g[f"o{lvl}_{i}"])), (lvl, i)
    assert torch.equal(od[3], torch.from_numpy(g["dual_v"])) and torch.equal(od[0][0], torch.from_numpy(g["dual_o04_0"]))
    ref_root = os.environ.get("OPENSTEREO_REF", "/root/reference")
    if os.path.isdir(ref_root):
        import importlib, sys, types
        from openstereo_amd import attach
        attach.stub_reference_packages(ref_root)
        sys.modules.setdefault("timm", types.ModuleType("timm"))
        Ref = importlib.import_module("stereo.modeling.models.igev.extractor").MultiBasicEncoder
        ref = Ref(output_dim=[HD, HD], norm_fn="batch", downsample=2)
        assert {k: tuple(v.shape) for k, v in ref.state_dict().items()} == {k: tuple(v.shape) for k, v in enc.state_dict().items()}


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_engine_path_vs_reference_golden(prec):
    """Eval mode on the GPU: every conv + BatchNorm (+ ReLU) one fused MFMA launch, the ResidualBlock sum relu(x + relu(.)) in the second
    conv's epilogue (OSA_RES_AFTER_ACT), the strided downsample branches, the five heads -- vs the reference's outputs, both modes."""
    from openstereo_amd import engine
    g = golden("context_encoder.npz")
    enc = _enc().cuda()
    img, _ = synth_images(2, 64, 128, seed=33, max_shift=8.0)
    old = engine.get_precision()
    engine.set_precision(prec)
    try:
        with torch.no_grad():
            o = enc(img.cuda(), num_layers=3)
            od = enc(img.cuda(), dual_inp=True, num_layers=3)
            o2 = enc(img.cuda(), num_layers=2)
    finally:
        engine.set_precision(old)
    assert len(o) == 3 and len(od) == 4 and len(o2) == 2
    for j, lvl in enumerate(("04", "08", "16")):
        for i in range(2):
            want = torch.from_numpy(g[f"o{lvl}_{i}"])
            torch.testing.assert_close(o[j][i].cpu(), want, rtol=2e-5, atol=2e-5 * max(1.0, float(want.abs().max())), msg=lambda s: f"outputs{lvl}[{i}] [{prec}]: {s}")
    torch.testing.assert_close(od[3].cpu(), torch.from_numpy(g["dual_v"]), rtol=2e-5, atol=5e-5)
    torch.testing.assert_close(od[0][0].cpu(), torch.from_numpy(g["dual_o04_0"]), rtol=2e-5, atol=5e-5)
    torch.testing.assert_close(o2[1][1], o[1][1], rtol=0, atol=0)


@pytest.mark.gpu
def test_training_mode_gradients_vs_torch():
    """Training mode (batch statistics) keeps the reference's torch composition with the eligible convolutions on the engine's autograd
    Functions: outputs and parameter gradients vs the same module run by stock PyTorch-ROCm.  ReLUs are smoothed in both runs
    (tests/_smooth.py: a gradient is discontinuous wherever a pre-activation is ~0, and two fp32 convolutions do not agree on the sign of
    a 1e-7 value).  Convolution biases in front of a batch-statistics BatchNorm have a mathematically ZERO gradient (the norm removes the
    mean): both runs return rounding noise there, so the absolute floor is tied to the largest gradient in the module."""
    import copy
    import contextlib
    from _smooth import smooth_activations
    from openstereo_amd import autograd as AG
    enc = _enc().cuda().train()
    ref = copy.deepcopy(enc)
    img, _ = synth_images(2, 32, 64, seed=34, max_shift=4.0)
    x = img.cuda()
    with smooth_activations():
        out = enc(x, num_layers=3)
        real, AG.engine_convs = AG.engine_convs, contextlib.nullcontext
        try:
            want = ref(x, num_layers=3)
        finally:
            AG.engine_convs = real
        loss = sum((a * a).mean() for lv in out for a in lv)
        loss_r = sum((a * a).mean() for lv in want for a in lv)
        loss.backward(); loss_r.backward()
    assert abs(float(loss) - float(loss_r)) < 1e-4 * abs(float(loss_r))
    pr = dict(ref.named_parameters())
    gmax = max(float(q.grad.abs().max()) for q in pr.values() if q.grad is not None)
    assert gmax > 0
    for k, p in enc.named_parameters():
        if p.grad is None:
            continue
        gr = pr[k].grad
        assert float((p.grad - gr).abs().max()) <= 1e-3 * float(gr.abs().max()) + 1e-5 * gmax, k
