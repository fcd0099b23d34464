"""CPU: the oracle restatement (oracle/torch_ref.py) against vectors produced by the REAL reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import torch

from conftest import golden
from oracle import torch_ref as O
from openstereo_amd.utils.weights import synth_state_dict, synth_images

T = torch.from_numpy


def close(a, b, atol=1e-6, rtol=1e-6):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def test_volumes_against_reference():
    g = golden("volumes.npz")
    for tag in ("a", "narrow", "k12"):
        B, C, H, W, D, G = g[f"{tag}_meta"]
        L, R = T(g[f"{tag}_L"]), T(g[f"{tag}_R"])
        close(O.gwc_volume(L, R, D, G), g[f"{tag}_gwc"])
        close(O.gwc_volume(L, R, D, G), g[f"{tag}_igev_gwc"])
        close(O.concat_volume(L, R, D), g[f"{tag}_concat"], 0, 0)
        close(O.concat_volume(L, R, D), g[f"{tag}_psm_cat"], 0, 0)
        close(O.concat_volume(L, R, D, mask_left=False), g[f"{tag}_igev_concat"], 0, 0)
        close(O.corr_volume(L, R, D), g[f"{tag}_corr"])
        close(O.build_corr_volume(L, R, D), g[f"{tag}_corr2"])
        fused = torch.cat((O.gwc_volume(L, R, D, G), O.concat_volume(L[:, :6], R[:, :6], D)), 1)
        close(fused, g[f"{tag}_gwcnet_volume"])


def test_regression_against_reference():
    g = golden("regression.npz")
    prob, cost = T(g["prob"]), T(g["cost"])
    close(O.disparity_regression(prob, 12, keepdim=True), g["reg_keep"])
    close(O.disparity_regression(prob, 12, keepdim=False), g["reg_nokeep"])
    close(O.softmax_regression(cost, keepdim=False), g["faster_softargmin"], atol=1e-5)
    close(O.upsample_regression(T(g["low"]), 24, 20, 28, False), g["up_false"], atol=1e-5)
    close(O.upsample_regression(T(g["low"]), 24, 20, 28, True), g["up_true"], atol=1e-5)
    close(O.upsample_regression(T(g["low2"]), 17, 13, 21, False), g["up_odd"], atol=1e-5)


def _shapes(prefix_model):
    return {k: tuple(v.shape) for k, v in prefix_model.state_dict().items()}


def test_gwc_hourglass_against_reference():
    from openstereo_amd.models.gwcnet import Hourglass
    g = golden("gwc_hourglass.npz")
    sd = {"hg." + k: v for k, v in synth_state_dict(Hourglass(8), seed=3).items()}
    close(O.gwc_hourglass(T(g["x"]), sd, "hg"), g["y"], atol=1e-5, rtol=1e-5)


def test_gwc_disp_processor_against_reference():
    from openstereo_amd.models.gwcnet import GwcDispProcessor
    g = golden("gwc_disp.npz")
    dp = GwcDispProcessor(maxdisp=32)
    sd = {"DispProcessor." + k: v for k, v in synth_state_dict(dp, seed=4).items()}
    taps = {}
    cost3 = O.gwc_aggregate(T(g["volume"]), sd, taps=taps)
    for k in ("cost0", "out1", "out3", "cost3"):
        close(taps[k], g[k], atol=2e-5, rtol=1e-5)
    close(O.upsample_regression(cost3, 32, 32, 64), g["disp"], atol=1e-4)


def test_gwcnet_small_against_reference():
    from openstereo_amd.models.gwcnet import GwcNet
    g = golden("gwcnet_small.npz")
    sd = synth_state_dict(GwcNet(), seed=0)
    L, R = synth_images(1, 64, 128, seed=1)
    taps = {}
    with torch.no_grad():
        disp = O.gwcnet_forward(L, R, sd, taps=taps)
    for k in ("left_gwc", "right_gwc", "left_cat", "right_cat"):
        close(taps[k], g[k], atol=1e-5, rtol=1e-5)
    close(taps["cost3"], g["cost3"], atol=1e-4, rtol=1e-4)
    epe = (disp.numpy() - g["disp"]).__abs__().mean()
    assert epe < 1e-4, epe


def test_psmnet_256x512_against_reference():
    """BASELINE configs[0]: PSMNet, one synthetic 256x512 pair, D=64, reference CPU path."""
    from openstereo_amd.models.psmnet import PSMNet, _Cfg
    g = golden("psmnet_256x512.npz")
    sd = synth_state_dict(PSMNet(_Cfg(MAX_DISP=64)), seed=0, head_gain=3.0)
    L, R = synth_images(1, 256, 512, seed=1, max_shift=16.0)
    taps = {}
    with torch.no_grad():
        d1, d2, d3 = O.psmnet_forward(L, R, sd, maxdisp=64, taps=taps)
    close(taps["left_feature"], g["left_feature"], atol=1e-5, rtol=1e-5)
    for d, k in ((d1, "disp1"), (d2, "disp2"), (d3, "disp3")):
        epe = np.abs(d.numpy() - g[k]).mean()
        assert epe < 1e-4, (k, epe)


def test_stereobase_and_igev_hourglass_against_reference():
    from openstereo_amd.models.igev_style import Hourglass, hourglass
    g = golden("stereobase_hourglass.npz")
    sd = {"h." + k: v for k, v in synth_state_dict(Hourglass(24, [96, 64, 192, 120]), seed=6).items()}
    f = [None, T(g["f1"]), T(g["f2"]), T(g["f3"])]
    y, y1, y2 = O.igev_style_hourglass(T(g["x"]), f, sd, "h", "stereobase", return_multi=True)
    close(y, g["y"], atol=2e-5, rtol=2e-5); close(y1, g["y1"], atol=2e-5, rtol=2e-5); close(y2, g["y2"], atol=2e-5, rtol=2e-5)
    g = golden("igev_hourglass.npz")
    sd = {"h." + k: v for k, v in synth_state_dict(hourglass(8), seed=7).items()}
    f = [None, T(g["f1"]), T(g["f2"]), T(g["f3"])]
    close(O.igev_style_hourglass(T(g["x"]), f, sd, "h", "igev"), g["y"], atol=2e-5, rtol=2e-5)


def test_context_upsample_against_reference():
    g = golden("context_upsample.npz")
    close(O.context_upsample(T(g["disp_low"]) * 4., T(g["weights"])), g["out"], atol=1e-6)
    close(O.context_upsample(T(g["disp_low"])[:, :, :3, :4], T(g["weights"])[:, :, :6, :8], 2), g["out_s2"], atol=1e-6)


def test_geo_encoding_volume_against_reference():
    g = golden("geo_encoding.npz")
    gev = O.GeoEncodingVolume(T(g["f1"]), T(g["f2"]), T(g["geo"]), 2, 4)
    close(gev(T(g["disp"]), T(g["coords"])), g["lookup"], atol=1e-6)
    close(gev(T(g["disp"]) * 2.5 + 1.0, T(g["coords"])), g["lookup2"], atol=1e-6)


def test_lightstereo_aggregation_against_reference():
    """a9: oracle restatement of LightStereo's Aggregation vs the real reference module's output."""
    from conftest import lightstereo_case
    agg, sd, x, feats = lightstereo_case()
    g = golden("lightstereo_agg.npz")
    # state_dict keys of the engine-side mirror are the reference's (checked when the fixture was made:
    # load_state_dict of the reference module with the same synth dict is strict)
    taps = {}
    with torch.no_grad():
        y = O.lightstereo_aggregation(x, feats, sd, taps=taps)
    for k, v in (("y", y), ("att0", taps["att0"]), ("att4", taps["att4"])):
        ref = torch.from_numpy(g[k])
        assert v.shape == ref.shape
        err = (v - ref).abs().max().item()
        assert err <= 2e-4 * max(1.0, ref.abs().max().item()), (k, err)


def test_igev_update_block_against_reference():
    """8f #4: oracle restatement of BasicMultiUpdateBlock (ConvGRUs, motion encoder, heads) vs the real reference."""
    from conftest import igev_update_case
    blk, sd, net, inp, corr, disp = igev_update_case()
    g = golden("igev_update.npz")
    with torch.no_grad():
        n, mask, delta = O.igev_update_block(net, inp, corr, disp, sd)
    for k, v in (("net0", n[0]), ("net1", n[1]), ("net2", n[2]), ("mask", mask), ("delta", delta)):
        ref = torch.from_numpy(g[k])
        assert v.shape == ref.shape
        assert (v - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()), k


def test_lightstereo_cost_stage_against_reference():
    """a4 -> a9 -> a12 chained as in lightstereo.py:51-56, oracle vs the real reference's functions."""
    from conftest import lightstereo_stage_case
    st, sd, fl, fr0 = lightstereo_stage_case()
    g = golden("lightstereo_stage.npz")
    with torch.no_grad():
        init, prob, enc = O.lightstereo_cost_stage(fl, fr0, sd, 192)
    assert (enc - torch.from_numpy(g["enc"])).abs().max().item() <= 2e-4 * max(1.0, float(np.abs(g["enc"]).max()))
    assert (init - torch.from_numpy(g["init_disp"])).abs().max().item() <= 1e-3


def test_igev_refine_loop_against_reference():
    """Three iterations of geometry lookup + slow-fast GRU updates + disparity update: oracle vs the reference's pieces."""
    from conftest import igev_refine_case
    _, sd, ml, mr, gvol, net, inp, d0 = igev_refine_case()
    g = golden("igev_refine.npz")
    with torch.no_grad():
        disp, mask, n = O.igev_refine(ml, mr, gvol, net, inp, d0, sd, 3)
    for k, v in (("disp", disp), ("mask", mask), ("net0", n[0])):
        ref = torch.from_numpy(g[k])
        assert (v - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), k


def test_at_size_fixtures_against_reference():
    """The oracle at BASELINE sizes (configs [2] and [4]): StereoBase cost stage at the 320x736 crop and the IGEV loop at 136x240
    with 32 iterations, vs the outputs of the real reference's modules (tests/golden/*_at_size.npz)."""
    from conftest import igev_at_size_case, stereobase_at_size_case
    g = golden("stereobase_at_size.npz")
    st, x, feats = stereobase_at_size_case()
    with torch.no_grad():
        d, prob, geo = O.stereobase_cost_stage(*x, feats, st.state_dict(), 192, 8)
    # (5e-5 px: the fixture was generated on another host CPU; oneDNN picks its convolution code path -- and with it the fp32 summation
    # order -- by ISA and thread count, which moves this soft-argmin by up to ~10 ulp of its ~25 px values)
    assert (d - torch.from_numpy(g["init_disp"])).abs().max().item() <= 5e-5
    assert (geo[:, :, ::4, ::4, ::4] - torch.from_numpy(g["geo_sub"])).abs().max().item() <= 1e-5 * float(np.abs(g["geo_sub"]).max())
    g = golden("igev_at_size.npz")
    ref, ml, mr, gvol, net, inp, d0 = igev_at_size_case()
    sd = {"update_block." + k: v for k, v in ref.update_block.state_dict().items()}
    with torch.no_grad():
        disp, mask, n = O.igev_refine(ml, mr, gvol, net, inp, d0, sd, 32)
    assert (disp - torch.from_numpy(g["disp"])).abs().max().item() <= 2e-4
    assert (n[0][:, :, ::4, ::4] - torch.from_numpy(g["net0_sub"])).abs().max().item() <= 2e-4


def test_preprocess_against_reference_transforms():
    """f3: the oracle's restatement of RightTopPad -> TransposeImage -> ToTensor -> NormalizeImage vs the output of the reference's own
    transform classes (stereo_trans.py; fixture generated with cv2 / torchvision stubbed, see make_golden.gen_preprocess)."""
    g = golden("preprocess.npz")
    for side in ("left", "right"):
        for img in (g[side + "_u8"], g[side + "_u8"].astype(np.float32)):
            out = O.preprocess_image(img, (32, 48))
            assert out.shape == (3, 32, 48)
            assert (out - torch.from_numpy(g[side])).abs().max().item() <= 1e-6


def test_dormant_volume_variants_oracle():
    """CoExCostVolume, compute_volume, build_sub_volume and cat_fms (negative start, dilation) restatements vs the reference's own outputs
    (dormant_volumes.npz; compute_volume / build_sub_volume were run with their hard-coded device='cuda' zeros redirected to the CPU,
    make_golden.gen_dormant), plus a tiny case written out by hand."""
    import torch
    from oracle import torch_ref as R
    g = golden("dormant_volumes.npz")
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    for grp in (1, 4):
        torch.testing.assert_close(R.coex_cost_volume(x, y, 6, grp), torch.from_numpy(g[f"coex_g{grp}"]), rtol=1e-5, atol=1e-6)
    assert torch.equal(R.compute_volume(x, y, 7, "left"), torch.from_numpy(g["compute_left"]))
    assert torch.equal(R.compute_volume(x, y, 7, "right"), torch.from_numpy(g["compute_right"]))
    torch.testing.assert_close(R.build_sub_volume(x, y, 7), torch.from_numpy(g["sub_volume"]), rtol=1e-6, atol=1e-6)
    for tag in ("neg", "dil", "negdil"):
        md, st, dil = (int(v) for v in g[f"catfms_{tag}_args"])
        assert torch.equal(R.cat_fms(x, y, md, st, dil), torch.from_numpy(g[f"catfms_{tag}"])), tag
    l = torch.tensor([[[[1.0, 2.0, 4.0]]], ]).repeat(1, 2, 1, 1); l[:, 1] *= -1          # [1,2,1,3]
    r = torch.tensor([[[[0.5, 1.0, 3.0]]], ]).repeat(1, 2, 1, 1)
    cv = R.compute_volume(l, r, 2, "left")
    assert cv.shape == (1, 2, 2, 1, 3) and cv[0, 0, 1, 0].tolist() == [0.0, 1.5, 3.0] and cv[0, 0, 0, 0].tolist() == [0.5, 1.0, 1.0]
    cvr = R.compute_volume(l, r, 2, "right")
    assert cvr[0, 0, 1, 0].tolist() == [0.0, 1.0, 0.0]                                   # target[w+1] - reference[w] for w < W-1
    sv = R.build_sub_volume(l, r, 2)
    assert sv.shape == (1, 2, 1, 3) and sv[0, 0, 0].tolist() == [2.0, 4.0, 8.0]          # |1-.5|+|-1-.5|, |2-1|+|-2-1|, |4-3|+|-4-3|
    assert sv[0, 1, 0].tolist() == [2.0, 4.0, 8.0]                                       # w<1: |1|+|-1|; |2-.5|+|-2-.5|; |4-1|+|-4-1|
