"""Generate the golden vectors under tests/golden/ by running the REAL OpenStereo reference.

Runs only in the build container (needs /root/reference; CPU torch).  The reference is imported
through stub parent packages so that stereo/modeling/__init__.py (which pulls cv2/timm/...) is
never executed.  Inputs and parameters are regenerated deterministically from seeds
by openstereo_amd.utils.weights, so fixtures store mostly *outputs*.

    python tests/golden/make_golden.py [--full]      # --full also writes the 544x960 GwcNet disparity
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("OPENSTEREO_REF", "/root/reference")
sys.path.insert(0, ROOT)


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not found: golden vectors can only be generated where the reference is mounted")
    sys.path.insert(0, REF)
    for name, path in [("stereo", "stereo"), ("stereo.modeling", "stereo/modeling"),
                       ("stereo.modeling.models", "stereo/modeling/models")]:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, path)]
        sys.modules[name] = m


def rnd(shape, seed):
    return torch.from_numpy(np.random.default_rng(seed).normal(0, 1, shape).astype(np.float32))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                      for k, v in arrays.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1e6:.2f} MB")


class Cfg(dict):
    __getattr__ = dict.__getitem__


def gen_lightstereo():
    """LightStereo-S aggregation (a9): cfgs/lightstereo/lightstereo_s_sceneflow.yaml -- in_channels 48,
    AGGREGATION_BLOCKS [1, 2, 4], EXPANSE_RATIO 4, LEFT_ATT true, MobileNetv2 channels [24, 32, 96, 160]."""
    from openstereo_amd.utils.weights import synth_state_dict
    from stereo.modeling.models.lightstereo.aggregation import Aggregation
    agg = Aggregation(in_channels=48, left_att=True, blocks=[1, 2, 4], expanse_ratio=4,
                      backbone_channels=[24, 32, 96, 160]).eval()
    agg.load_state_dict(synth_state_dict(agg, seed=9))
    x = rnd((1, 48, 32, 64), 51)
    feats = [rnd((1, 24, 32, 64), 52), rnd((1, 32, 16, 32), 53), rnd((1, 96, 8, 16), 54), rnd((1, 160, 4, 8), 55)]
    taps = {}
    agg.att0.register_forward_hook(lambda m, i, o: taps.__setitem__("att0", o.clone()))
    agg.att4.register_forward_hook(lambda m, i, o: taps.__setitem__("att4", o.clone()))
    y = agg(x, feats)[0]
    print("LightStereo aggregation out range", y.min().item(), y.max().item(), y.std().item())
    save("lightstereo_agg.npz", y=y, **taps)      # inputs: rnd(shape, 51..54), see tests/conftest.py lightstereo_inputs()
    # cost stage of lightstereo.py:51-56 from the reference's own functions (the LightStereo class itself needs timm)
    import torch.nn.functional as F
    from stereo.modeling.cost_volume.cost_volume import correlation_volume
    from stereo.modeling.disp_pred.disp_regression import disparity_regression
    fl = [rnd((1, 24, 32, 64), 56)] + feats[1:]
    fr0 = torch.roll(fl[0], shifts=-3, dims=3) + 0.1 * rnd((1, 24, 32, 64), 57)
    vol = correlation_volume(fl[0], fr0, 48)
    enc = agg(vol, fl)[0]
    init = disparity_regression(F.softmax(enc, dim=1), 48)
    print("LightStereo cost stage: init_disp range", init.min().item(), init.max().item())
    save("lightstereo_stage.npz", init_disp=init, enc=enc)


def gen_igev_update():
    """IGEV BasicMultiUpdateBlock (8f #4): cfgs/igev defaults -- CORR_LEVELS 2, CORR_RADIUS 4, N_GRU_LAYERS 3,
    N_DOWNSAMPLE 2, HIDDEN_DIMS [128, 128, 128]; one full iteration (all three GRUs + heads)."""
    import importlib.util
    from openstereo_amd.utils.weights import synth_state_dict
    spec = importlib.util.spec_from_file_location("ref_igev_update", os.path.join(REF, "stereo/modeling/models/igev/update.py"))
    upd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(upd)
    args = Cfg(CORR_LEVELS=2, CORR_RADIUS=4, N_GRU_LAYERS=3, N_DOWNSAMPLE=2)
    blk = upd.BasicMultiUpdateBlock(args, hidden_dims=[128, 128, 128]).eval()
    blk.load_state_dict(synth_state_dict(blk, seed=11))
    H, W = 16, 32
    net = [torch.tanh(rnd((1, 128, H >> i, W >> i), 70 + i)) for i in range(3)]
    inp = [[rnd((1, 128, H >> i, W >> i), 80 + 3 * i + j) * 0.5 for j in range(3)] for i in range(3)]
    corr, disp = rnd((1, 162, H, W), 90), rnd((1, 1, H, W), 91).abs() * 10
    n, mask, delta = blk([t.clone() for t in net], inp, corr, disp)
    print("IGEV update block: delta range", delta.min().item(), delta.max().item())
    save("igev_update.npz", net0=n[0], net1=n[1], net2=n[2], mask=mask, delta=delta)   # inputs: rnd(...) as in tests/conftest.py igev_update_case()
    # ---- three iterations of the refinement loop of igev_stereo.py:181-203 built from the reference's own pieces
    sys.modules.setdefault("timm", types.ModuleType("timm"))
    from stereo.modeling.models.igev.geometry import Combined_Geo_Encoding_Volume
    ml, mr = rnd((1, 96, H, W), 92), rnd((1, 96, H, W), 93)
    gvol = rnd((1, 8, 12, H, W), 94)
    geo_fn = Combined_Geo_Encoding_Volume(ml, mr, gvol, radius=4, num_levels=2)
    coords = torch.arange(W).float().reshape(1, 1, W, 1).repeat(1, H, 1, 1)
    d = rnd((1, 1, H, W), 95).abs() * 3
    nl = [t.clone() for t in net]
    for _ in range(3):
        gf = geo_fn(d, coords)
        nl = blk(nl, inp, iter16=True, iter08=False, iter04=False, update=False)
        nl = blk(nl, inp, iter16=True, iter08=True, iter04=False, update=False)
        nl, mk, dd = blk(nl, inp, gf, d, iter16=True, iter08=True)
        d = d + dd
    print("IGEV refine loop: disp range", d.min().item(), d.max().item())
    save("igev_refine.npz", disp=d, mask=mk, net0=nl[0])


def gen_at_size():
    """BASELINE configs [2] and [4] at their real sizes, sub-sampled taps so the fixtures stay small:
      * geometry-encoding lookup at [1,96,136,240] features / [1,8,48,136,240] volume (igev/geometry.py:7-66),
      * the IGEV refinement loop at 136x240 with VALID_ITERS = 32 (igev_stereo.py:181-203, cfgs/igev/igev_sceneflow_amp.yaml:31),
      * the StereoBase cost stage at the 320x736 training crop (quarter resolution 80x184, D/4 = 48; stereobase_gru.py:139-164).
    Inputs are regenerated from seeds by tests/conftest.py (igev_at_size_case / stereobase_at_size_case).
    GRU weights use gain 0.8: with the unit-gain synthetic weights the 32-step recurrence amplifies a 1e-6 perturbation 800x
    (chaotic regime, measured with the oracle), which would turn a kernel comparison into a lottery; at 0.8 the map is
    contractive (amplification ~7x), disparities still move by +-17 px over the 32 iterations."""
    import importlib.util
    import torch.nn.functional as F
    from openstereo_amd.utils.weights import synth_state_dict
    sys.modules.setdefault("timm", types.ModuleType("timm"))
    from stereo.modeling.models.igev.geometry import Combined_Geo_Encoding_Volume
    spec = importlib.util.spec_from_file_location("ref_igev_update", os.path.join(REF, "stereo/modeling/models/igev/update.py"))
    upd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(upd)
    H, W = 136, 240
    ml, mr = rnd((1, 96, H, W), 192), rnd((1, 96, H, W), 193)
    gvol = rnd((1, 8, 48, H, W), 194)
    coords = torch.arange(W).float().reshape(1, 1, W, 1).repeat(1, H, 1, 1)
    d0 = rnd((1, 1, H, W), 195).abs() * 3
    geo_fn = Combined_Geo_Encoding_Volume(ml, mr, gvol, radius=4, num_levels=2)
    lk = geo_fn(d0 * 8.0, coords)                                        # disparities up to ~100: taps leave the volume on both levels
    args = Cfg(CORR_LEVELS=2, CORR_RADIUS=4, N_GRU_LAYERS=3, N_DOWNSAMPLE=2)
    blk = upd.BasicMultiUpdateBlock(args, hidden_dims=[128, 128, 128]).eval()
    blk.load_state_dict(synth_state_dict(blk, seed=11, gain=0.8))
    net = [torch.tanh(rnd((1, 128, H >> i, W >> i), 170 + i)) for i in range(3)]
    inp = [[rnd((1, 128, H >> i, W >> i), 180 + 3 * i + j) * 0.5 for j in range(3)] for i in range(3)]
    d, nl, t = d0, [x.clone() for x in net], time.time()
    trace = []
    for it in range(32):
        gf = geo_fn(d, coords)
        nl = blk(nl, inp, iter16=True, iter08=False, iter04=False, update=False)
        nl = blk(nl, inp, iter16=True, iter08=True, iter04=False, update=False)
        nl, mk, dd = blk(nl, inp, gf, d, iter16=True, iter08=True)
        d = d + dd
        if it in (0, 7, 15):
            trace.append(d[:, :, ::4, ::4].clone())
    print(f"IGEV refine x32 at {H}x{W}: {time.time() - t:.1f} s, disp range {d.min().item():.2f}..{d.max().item():.2f}")
    save("igev_at_size.npz", lookup_sub=lk[:, :, ::8, ::8], disp=d, disp_it1=trace[0], disp_it8=trace[1], disp_it16=trace[2],
         mask_sub=mk[:, :, ::4, ::4], net0_sub=nl[0][:, :, ::4, ::4])

    # ---- StereoBase cost stage at the training crop
    from stereo.modeling.cost_volume.cost_volume import build_gwc_volume, build_concat_volume
    from stereo.modeling.disp_pred.disp_regression import disparity_regression
    from stereo.modeling.models.stereobase.hourglass import Hourglass as SBHourglass
    from openstereo_amd.models.igev_style import StereoBaseCostStage
    h, w = 80, 184
    st = StereoBaseCostStage(max_disp=192, num_groups=8, concat_channels=8, backbone_channels=[48, 64, 192, 120])
    sd = synth_state_dict(st, seed=8, head_gain=20.0)
    hg = SBHourglass(24, [48, 64, 192, 120]).eval()
    hg.load_state_dict({k[len("cost_agg."):]: v for k, v in sd.items() if k.startswith("cost_agg.")})
    fm_l, fm_r = rnd((1, 96, h, w), 201), rnd((1, 96, h, w), 202)
    ct_l, ct_r = rnd((1, 8, h, w), 203), rnd((1, 8, h, w), 204)
    feats = [None, rnd((1, 64, h // 2, w // 2), 205), rnd((1, 192, h // 4, w // 4), 206), rnd((1, 120, h // 8, w // 8), 207)]
    t = time.time()
    vol = torch.cat((build_gwc_volume(fm_l, fm_r, 48, 8), build_concat_volume(ct_l, ct_r, 48)), 1)     # stereobase_gru.py:156-160
    geo = hg(vol, feats)
    cost = F.conv3d(geo, sd["classifier.weight"], None, 1, 1).squeeze(1)                              # :162
    prob = F.softmax(cost, dim=1)
    init_disp = disparity_regression(prob, 48)                                                         # :163-164
    print(f"StereoBase cost stage at {h}x{w}: {time.time() - t:.1f} s, init disp range {init_disp.min().item():.2f}..{init_disp.max().item():.2f}"
          f" std {init_disp.std().item():.2f}")
    save("stereobase_at_size.npz", init_disp=init_disp, prob_sub=prob[:, :, ::4, ::4], geo_sub=geo[:, :, ::4, ::4, ::4])


def gen_preprocess():
    """Input pre-processing (8f #3): the reference's own transform classes, stereo/datasets/dataset_utils/stereo_trans.py --
    RightTopPad (:243-267) -> TransposeImage (:22-29) -> ToTensor (:32-44) -> NormalizeImage (:48-56), composed as
    cfgs/gwcnet/gwcnet_sceneflow.yaml:13-19 does for evaluation.  The module imports cv2 and torchvision at the top; neither is
    installed here and only `torchvision.transforms.functional.normalize` is exercised by this chain, so both are stubbed and
    normalize is restated with torchvision's semantics (float tensor, out = (x - mean[:, None, None]) / std[:, None, None])."""
    import importlib.util

    def tv_normalize(tensor, mean, std, inplace=False):
        t = tensor.clone()
        m = torch.as_tensor(mean, dtype=t.dtype).view(-1, 1, 1)
        sd = torch.as_tensor(std, dtype=t.dtype).view(-1, 1, 1)
        return t.sub_(m).div_(sd)
    tv, tvt, tvf = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.transforms.functional")
    tvf.normalize = tv_normalize
    tvt.functional, tvt.ColorJitter = tvf, object
    tv.transforms = tvt
    for name, mod in (("cv2", types.ModuleType("cv2")), ("torchvision", tv), ("torchvision.transforms", tvt),
                      ("torchvision.transforms.functional", tvf)):
        sys.modules.setdefault(name, mod)
    spec = importlib.util.spec_from_file_location("ref_stereo_trans", os.path.join(REF, "stereo/datasets/dataset_utils/stereo_trans.py"))
    st = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(st)
    chain = st.Compose([st.RightTopPad(Cfg(SIZE=[32, 48])), st.TransposeImage(Cfg()), st.ToTensor(Cfg()),
                        st.NormalizeImage(Cfg(MEAN=[0.485, 0.456, 0.406], STD=[0.229, 0.224, 0.225]))])
    r = np.random.default_rng(12)
    L = r.integers(0, 256, (27, 45, 3)).astype(np.uint8)
    R = r.integers(0, 256, (27, 45, 3)).astype(np.uint8)
    out = chain({"left": L.astype(np.float32), "right": R.astype(np.float32)})      # dataset readers hand float32 HWC arrays over
    assert tuple(out["left"].shape) == (3, 32, 48)
    save("preprocess.npz", left_u8=L, right_u8=R, left=out["left"], right=out["right"])


def gen_dormant():
    """Dormant volume variants.  CoExCostVolume (cost_volume.py:9-29) is CPU-runnable.  compute_volume / build_sub_volume (:44-56, :108-117)
    hard-code device='cuda' in their torch.zeros calls: the module's `torch` global is replaced by a proxy whose zeros() drops that
    keyword, everything else is the reference's code.  cat_fms (psmnet_cost_processor.py:9-50) with negative start
    disparities and dilation."""
    from stereo.modeling.cost_volume import cost_volume as cv
    from stereo.modeling.models.psmnet.psmnet_cost_processor import cat_fms

    class _CpuTorch:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def zeros(*a, **k):
            k.pop("device", None)
            return torch.zeros(*a, **k)
    x, y = rnd((2, 8, 5, 20), 301), rnd((2, 8, 5, 20), 302)
    out = {"x": x.numpy(), "y": y.numpy()}
    for grp in (1, 4):
        out[f"coex_g{grp}"] = cv.CoExCostVolume(6, grp)(x, y).numpy()
    real = cv.torch
    cv.torch = _CpuTorch()
    try:
        out["compute_left"] = cv.compute_volume(x, y, 7, "left").numpy()
        out["compute_right"] = cv.compute_volume(x, y, 7, "right").numpy()
        out["sub_volume"] = cv.build_sub_volume(x, y, 7).numpy()
    finally:
        cv.torch = real
    for tag, (md, st, dil) in {"neg": (9, -4, 1), "dil": (12, 2, 3), "negdil": (10, -6, 2)}.items():
        out[f"catfms_{tag}"] = cat_fms(x, y, max_disp=md, start_disp=st, dilation=dil).numpy()
        out[f"catfms_{tag}_args"] = np.array([md, st, dil])
    save("dormant_volumes.npz", **out)


def gen_unit_gain():
    """IGEV refinement x32 at 136x240 with UNIT-gain update-block weights (gen_at_size uses gain 0.8 because the recurrence amplifies a
    1e-6 perturbation ~800x at unit gain).  Stored next to the result: how far the reference's OWN output moves when the initial hidden
    state is perturbed by 1e-6 (relative) -- the yardstick an fp32-class implementation is held to."""
    import importlib.util
    from openstereo_amd.utils.weights import synth_state_dict
    sys.modules.setdefault("timm", types.ModuleType("timm"))
    from stereo.modeling.models.igev.geometry import Combined_Geo_Encoding_Volume
    spec = importlib.util.spec_from_file_location("ref_igev_update", os.path.join(REF, "stereo/modeling/models/igev/update.py"))
    upd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(upd)
    H, W = 136, 240
    ml, mr = rnd((1, 96, H, W), 192), rnd((1, 96, H, W), 193)
    gvol = rnd((1, 8, 48, H, W), 194)
    coords = torch.arange(W).float().reshape(1, 1, W, 1).repeat(1, H, 1, 1)
    d0 = rnd((1, 1, H, W), 195).abs() * 3
    geo_fn = Combined_Geo_Encoding_Volume(ml, mr, gvol, radius=4, num_levels=2)
    args = Cfg(CORR_LEVELS=2, CORR_RADIUS=4, N_GRU_LAYERS=3, N_DOWNSAMPLE=2)
    blk = upd.BasicMultiUpdateBlock(args, hidden_dims=[128, 128, 128]).eval()
    blk.load_state_dict(synth_state_dict(blk, seed=11))                      # default gain: the unit-gain recurrence
    net = [torch.tanh(rnd((1, 128, H >> i, W >> i), 170 + i)) for i in range(3)]
    inp = [[rnd((1, 128, H >> i, W >> i), 180 + 3 * i + j) * 0.5 for j in range(3)] for i in range(3)]

    def run(nl):
        d, nl = d0, [x.clone() for x in nl]
        for _ in range(32):
            gf = geo_fn(d, coords)
            nl = blk(nl, inp, iter16=True, iter08=False, iter04=False, update=False)
            nl = blk(nl, inp, iter16=True, iter08=True, iter04=False, update=False)
            nl, mk, dd = blk(nl, inp, gf, d, iter16=True, iter08=True)
            d = d + dd
        return d
    t = time.time()
    d_ref = run(net)
    d_pert = run([net[0] * (1.0 + 1e-6)] + net[1:])
    dev = (d_pert - d_ref).abs()
    print(f"IGEV refine x32 unit gain: {time.time() - t:.1f} s, disp range {d_ref.min().item():.2f}..{d_ref.max().item():.2f}; a 1e-6 relative "
          f"perturbation of net[0] moves the result by mean {dev.mean().item():.2e} / max {dev.max().item():.2e} px")
    save("igev_unit_gain.npz", disp=d_ref, pert_mean=dev.mean(), pert_max=dev.max(), pert_p999=torch.quantile(dev.flatten(), 0.999))


def gen_context_encoder():
    """MultiBasicEncoder (models/igev/extractor.py:194-297; the StereoBase copy models/stereobase/gru_blocks.py:62-148 must agree): the
    context network in front of the GRU loop, plain PyTorch in the reference (its file only imports timm) -> pinned against the reference's
    own class, norm_fn='batch', downsample=2 (cfgs/igev, cfgs/stereobase), eval mode, 64x128 image."""
    from openstereo_amd.utils.weights import synth_state_dict, synth_images
    sys.modules.setdefault("timm", types.ModuleType("timm"))
    from stereo.modeling.models.igev.extractor import MultiBasicEncoder as IGEVEnc
    from stereo.modeling.models.stereobase.gru_blocks import MultiBasicEncoder as SBEnc
    hd = [128, 128, 128]
    enc = IGEVEnc(output_dim=[hd, hd], norm_fn="batch", downsample=2).eval()
    sd = synth_state_dict(enc, seed=19, gain=0.9)
    enc.load_state_dict(sd)
    sb = SBEnc(output_dim=[hd, hd], norm_fn="batch", downsample=2).eval()
    assert set(sb.state_dict()) == set(sd)
    sb.load_state_dict(sd)
    img, _ = synth_images(2, 64, 128, seed=33, max_shift=8.0)
    o = enc(img, num_layers=3)
    o2 = sb(img, num_layers=3)
    for a, b in zip(o, o2):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    od = enc(img, dual_inp=True, num_layers=3)
    out = {f"o{lvl}_{i}": o[j][i] for j, lvl in enumerate(("04", "08", "16")) for i in range(2)}
    out["dual_v"] = od[3]
    out["dual_o04_0"] = od[0][0]
    print("MultiBasicEncoder: |out04| max", o[0][0].abs().max().item(), "std", o[0][0].std().item())
    save("context_encoder.npz", **out)


def _reference_model(modname, clsname, cfg, stand_ins):
    """Construct the reference's OWN model class with its timm-backed feature pyramid (`Feature` / `Backbone`: SURVEY 8 "out of scope",
    timm + pretrained weights are not available offline) replaced by the stand-in module the engine classes inject by default
    (openstereo_amd.models.stereo_models.StubFeature: a plain torch conv pyramid).  Everything else -- __init__, forward, the context
    network (MultiBasicEncoder), the 2-D heads, the iteration schedule, the final upsampling -- is the reference's code."""
    import importlib
    sys.modules.setdefault("timm", types.ModuleType("timm"))
    mod = importlib.import_module(modname)
    saved = {n: getattr(mod, n) for n in stand_ins}
    try:
        for n, f in stand_ins.items():
            setattr(mod, n, f)
        return getattr(mod, clsname)(cfg).eval()
    finally:
        for n, v in saved.items():
            setattr(mod, n, v)


def gen_e2e(full=True):
    """Whole-model forwards of the reference's StereoBase / IGEVStereo / LightStereo classes (stereobase_gru.py:121-213,
    igev_stereo.py:139-218, lightstereo.py:44-71) -- VERDICT r2 row h: the end-to-end classes of openstereo_amd/models/stereo_models.py
    are pinned against THESE outputs, not against an assembly of oracle stages.  128x256, MAX_DISP 64, 4 GRU iterations; plus IGEV-Stereo
    at 544x960, MAX_DISP 192, 32 iterations (contractive update-block weights as in gen_at_size), sub-sampled.
    Inputs / parameters: synth_images(seed=31) and synth_state_dict(seed, head_gain=20, gain=0.9), regenerated by the tests."""
    from openstereo_amd.utils.weights import synth_state_dict, synth_images
    from openstereo_amd.models.stereo_models import StubFeature
    H, W, MAXD = 128, 256, 64
    feat = lambda *a, **k: StubFeature((48, 64, 192, 160))
    cnet = None            # r3: MultiBasicEncoder is the reference's own class on both sides (only the timm pyramid is a stand-in)
    L, R = synth_images(1, H, W, seed=31, max_shift=12.0)
    out = {}

    def load(net, seed, gru_gain=None):
        sd = synth_state_dict(net, seed=seed, head_gain=20.0, gain=0.9)
        if gru_gain is not None:
            sd.update({k: v for k, v in synth_state_dict(net, seed=seed, head_gain=20.0, gain=gru_gain).items() if k.startswith("update_block.")})
        net.load_state_dict(sd)

    cfg = Cfg(MAX_DISP=MAXD, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, USE_GWC_VOLUME=True, USE_SUB_VOLUME=False, USE_INTERLACED_VOLUME=False,
              CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128], N_DOWNSAMPLE=2, N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2,
              SLOW_FAST_GRU=False, TRAIN_ITERS=4, EVAL_ITERS=4)
    sb = _reference_model("stereo.modeling.models.stereobase.stereobase_gru", "StereoBase", cfg, {"Feature": feat})
    load(sb, 41)
    r = sb({"left": L, "right": R})
    assert len(r["disp_preds"]) == 4
    out.update(stereobase_disp=r["disp_pred"], stereobase_init=r["init_disp"], stereobase_it1=r["disp_preds"][0])
    print("StereoBase e2e: disp range", r["disp_pred"].min().item(), r["disp_pred"].max().item(), "std", r["disp_pred"].std().item())
    # slow-fast schedule too (cfgs/stereobase: SLOW_FAST_GRU false; the code path exists, stereobase_gru.py:184-196)
    cfg2 = Cfg(cfg, SLOW_FAST_GRU=True)
    sb2 = _reference_model("stereo.modeling.models.stereobase.stereobase_gru", "StereoBase", cfg2, {"Feature": feat})
    load(sb2, 41)
    out["stereobase_slowfast_disp"] = sb2({"left": L, "right": R})["disp_pred"]

    args = Cfg(MAX_DISP=MAXD, HIDDEN_DIMS=[128, 128, 128], N_DOWNSAMPLE=2, N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2,
               SLOW_FAST_GRU=True, VALID_ITERS=4, TRAIN_ITERS=4)
    ig = _reference_model("stereo.modeling.models.igev.igev_stereo", "IGEVStereo", args, {"Feature": feat})
    load(ig, 43)
    L255, R255 = (L * 40 + 128).clamp(0, 255), (R * 40 + 128).clamp(0, 255)
    r = ig({"left": L255, "right": R255})
    out["igev_disp"] = r["disp_pred"]
    print("IGEV e2e: disp range", r["disp_pred"].min().item(), r["disp_pred"].max().item(), "std", r["disp_pred"].std().item())

    # the reference hard-codes the aggregation's in_channels = 48 = 192 / 4 (lightstereo.py:23): MAX_DISP must be 192
    lcfg = Cfg(MAX_DISP=192, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)
    ls = _reference_model("stereo.modeling.models.lightstereo.lightstereo", "LightStereo", lcfg,
                          {"Backbone": lambda *a, **k: StubFeature((24, 32, 96, 160))})
    load(ls, 47)
    r = ls({"left": L, "right": R})
    out["lightstereo_disp"] = r["disp_pred"]
    print("LightStereo e2e: disp range", r["disp_pred"].min().item(), r["disp_pred"].max().item(), "std", r["disp_pred"].std().item())

    if full:
        # BASELINE configs[4] at size: cfgs/igev/igev_sceneflow_amp.yaml (MAX_DISP 192, VALID_ITERS 32), 544x960 (540 padded to /32)
        args = Cfg(args, MAX_DISP=192, VALID_ITERS=32)
        ig = _reference_model("stereo.modeling.models.igev.igev_stereo", "IGEVStereo", args, {"Feature": feat})
        load(ig, 43, gru_gain=0.8)
        Lf, Rf = synth_images(1, 544, 960, seed=31)
        t = time.time()
        r = ig({"left": (Lf * 40 + 128).clamp(0, 255), "right": (Rf * 40 + 128).clamp(0, 255)})
        print(f"IGEV e2e 544x960 x32: {time.time() - t:.1f} s, disp range {r['disp_pred'].min().item():.2f}..{r['disp_pred'].max().item():.2f}"
              f" std {r['disp_pred'].std().item():.2f}")
        out["igev_full_disp_sub"] = r["disp_pred"][:, :, ::4, ::4]
    save("e2e_reference.npz", **out)
    gen_e2e_train(feat, cnet)


E2E_TRAIN_KEYS = {
    "stereobase": ("classifier.weight", "cost_agg.conv1.0.block.0.weight", "cost_agg.conv3_up.block.0.weight", "cost_agg.agg_0.1.block.0.weight",
                   "update_block.gru04.convz.weight", "update_block.gru16.convq.weight", "update_block.disp_head.conv2.weight",
                   "update_block.encoder.convc1.weight", "desc.weight", "concat_conv.1.weight", "spx_gru.0.weight", "spx.0.weight",
                   "context_zqr_convs.0.weight", "feature.stem.0.0.weight", "cnet.layer3.1.conv2.weight", "cnet.outputs08.1.0.conv1.weight"),
    "igev": ("classifier.weight", "corr_stem.conv.weight", "cost_agg.conv1.0.conv.weight", "cost_agg.conv2_up.conv.weight",
             "corr_feature_att.feat_att.1.weight", "update_block.gru08.convr.weight", "update_block.gru16.convq.weight",
             "update_block.disp_head.conv2.weight", "update_block.mask_feat_4.0.weight", "desc.weight", "conv.conv.weight",
             "spx_2_gru.conv1.conv.weight", "spx_gru.0.weight", "spx.0.weight", "stem_2.0.conv.weight", "feature.stem.0.0.weight",
             "cnet.conv1.weight", "cnet.layer2.1.conv1.weight", "cnet.outputs16.0.weight"),
    "lightstereo": ("cost_agg.conv0.0.pwconv.0.weight", "cost_agg.conv6.0.weight", "cost_agg.conv1.dwconv.0.weight", "cost_agg.att0.conv1_2.weight",
                    "refine_3.block.0.weight", "refine_1.0.block.0.weight", "stem_2.0.block.0.weight", "backbone.stem.0.0.weight"),
}


def gen_e2e_train(feat, cnet):
    """Training-mode forward + the reference's own get_loss + CPU autograd of the reference's model classes (frozen BatchNorm, i.e. the
    `freeze_bn` semantics of igev_stereo.py:121-124 applied to every BN so the fixture does not depend on batch statistics of one image):
    loss and the gradients of selected parameters across all stages (the end-to-end training tests used to assert only
    that the loss goes down).  64x128, 3 GRU iterations."""
    from openstereo_amd.utils.weights import synth_state_dict, synth_images
    from openstereo_amd.models.stereo_models import StubFeature
    import torch.nn as nn
    H, W = 64, 128
    L, R = synth_images(1, H, W, seed=31, max_shift=12.0)
    gt = torch.from_numpy(np.random.default_rng(3).uniform(1.0, 30.0, (1, H, W)).astype(np.float32))
    out = {}

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from _smooth import smooth_activations
    import contextlib

    def run(tag, net, seed, left, right):
        """Two fixtures per class: the reference as written (`<tag>_*`), and the reference with the ReLU-family kinks smoothed
        (`<tag>_smooth_*`, tests/_smooth.py: the whole-model gradient is discontinuous at every ~0 pre-activation, so only the smoothed
        network can be pinned tightly across implementations)."""
        net.load_state_dict(synth_state_dict(net, seed=seed, head_gain=20.0, gain=0.9))
        net.train()
        for m in net.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                m.eval()
        for sfx, ctx in (("", contextlib.nullcontext), ("_smooth", smooth_activations)):
            net.zero_grad(set_to_none=True)
            with torch.enable_grad(), ctx():
                pred = net({"left": left, "right": right})
                loss, _ = net.get_loss(pred, {"disp": gt})
                loss.backward()
            params = dict(net.named_parameters())
            out[f"{tag}{sfx}_loss"] = loss.detach()
            out[f"{tag}{sfx}_disp"] = pred["disp_pred"].detach()
            for k in E2E_TRAIN_KEYS[tag]:
                g = params[k].grad
                assert g is not None and float(g.abs().max()) > 0, (tag, k)
                out[f"{tag}{sfx}_grad::{k}"] = g.reshape(-1)[:20000].clone()
            print(f"{tag}{sfx} training step (reference autograd): loss {loss.item():.4f}")

    cfg = Cfg(MAX_DISP=64, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, USE_GWC_VOLUME=True, USE_SUB_VOLUME=False, USE_INTERLACED_VOLUME=False,
              CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128], N_DOWNSAMPLE=2, N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2,
              SLOW_FAST_GRU=False, TRAIN_ITERS=3, EVAL_ITERS=4)
    run("stereobase", _reference_model("stereo.modeling.models.stereobase.stereobase_gru", "StereoBase", cfg,
                                       {"Feature": feat}), 41, L, R)
    args = Cfg(MAX_DISP=64, HIDDEN_DIMS=[128, 128, 128], N_DOWNSAMPLE=2, N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2,
               SLOW_FAST_GRU=True, VALID_ITERS=4, TRAIN_ITERS=3)
    run("igev", _reference_model("stereo.modeling.models.igev.igev_stereo", "IGEVStereo", args, {"Feature": feat}),
        43, (L * 40 + 128).clamp(0, 255), (R * 40 + 128).clamp(0, 255))
    lcfg = Cfg(MAX_DISP=192, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)
    run("lightstereo", _reference_model("stereo.modeling.models.lightstereo.lightstereo", "LightStereo", lcfg,
                                        {"Backbone": lambda *a, **k: StubFeature((24, 32, 96, 160))}), 47, L, R)
    save("e2e_reference_train.npz", **out)


def gen_e2e_train_at_size():
    """BASELINE configs[2] AT SIZE: the reference's own StereoBase class, training mode, at the SceneFlow crop 320x736
    with MAX_DISP 192 and TRAIN_ITERS 22 (cfgs/stereobase/stereobase_sceneflow.yaml:15-16,27,40): forward, the reference's get_loss and
    CPU autograd, frozen BatchNorm.  Smoothed activations (tests/_smooth.py) like the tight half of gen_e2e_train -- a 22-iteration
    gradient through real ReLUs has thousands of ~0 pre-activations and cannot be pinned across implementations -- and contractive
    update-block weights (gain 0.8, as the at-size inference fixtures).  Stored: loss, final / first-iteration / initial disparities
    (sub-sampled), and the first 20000 elements of the gradients of parameters across all stages."""
    from openstereo_amd.utils.weights import synth_state_dict, synth_images
    from openstereo_amd.models.stereo_models import StubFeature
    import torch.nn as nn
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from _smooth import smooth_activations
    H, W = 320, 736
    feat = lambda *a, **k: StubFeature((48, 64, 192, 160))
    L, R = synth_images(1, H, W, seed=33, max_shift=40.0)
    gt = torch.from_numpy(np.random.default_rng(5).uniform(1.0, 120.0, (1, H, W)).astype(np.float32))
    cfg = Cfg(MAX_DISP=192, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, USE_GWC_VOLUME=True, USE_SUB_VOLUME=False, USE_INTERLACED_VOLUME=False,
              CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128], N_DOWNSAMPLE=2, N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2,
              SLOW_FAST_GRU=False, TRAIN_ITERS=22, EVAL_ITERS=32)
    net = _reference_model("stereo.modeling.models.stereobase.stereobase_gru", "StereoBase", cfg, {"Feature": feat})
    sd = synth_state_dict(net, seed=41, head_gain=20.0, gain=0.9)
    sd.update({k: v for k, v in synth_state_dict(net, seed=41, head_gain=20.0, gain=0.8).items() if k.startswith("update_block.")})
    net.load_state_dict(sd)
    net.train()
    for m in net.modules():
        if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
            m.eval()
    t = time.time()
    with torch.enable_grad(), smooth_activations():
        pred = net({"left": L, "right": R})
        loss, _ = net.get_loss(pred, {"disp": gt})
        loss.backward()
    print(f"StereoBase 320x736 x22 training step (reference autograd, smoothed activations): {time.time() - t:.1f} s, loss {loss.item():.4f}, "
          f"disp {pred['disp_pred'].min().item():.2f}..{pred['disp_pred'].max().item():.2f}")
    out = {"loss": loss.detach(), "disp_sub": pred["disp_pred"].detach()[..., ::4, ::4], "init_disp_sub": pred["init_disp"].detach()[..., ::2, ::2],
           "it1_sub": pred["disp_preds"][0].detach()[..., ::4, ::4], "n_preds": torch.tensor(len(pred["disp_preds"]))}
    params = dict(net.named_parameters())
    for k in E2E_TRAIN_KEYS["stereobase"]:
        g = params[k].grad
        assert g is not None and float(g.abs().max()) > 0, k
        out[f"grad::{k}"] = g.reshape(-1)[:20000].clone()
        out[f"gmax::{k}"] = g.abs().max()
    save("e2e_reference_train_at_size.npz", **out)


def gen_feature_pyramid():
    """The reference's OWN feature-pyramid classes -- `Feature` of StereoBase (models/stereobase/backbone.py:32-73) and IGEV-Stereo
    (models/igev/extractor.py:320-355), `Backbone` of LightStereo (models/lightstereo/backbone.py:29-75) -- with `timm.create_model`
    answered by openstereo_amd.models.feature_pyramid.create_model (the key-compatible MobileNetV2-100 mirror; timm itself is not available
    offline: the TRUNK is therefore unpinned, the FPN decoders -- Conv2xUp / Conv2x_IN / FPNLayer, InstanceNorm, replicate-padded
    out_conv -- and the forward wiring are the reference's code).  128x256 image; outputs at 1/4 .. 1/32 and the state_dict key list."""
    import importlib
    from openstereo_amd.models import feature_pyramid as FP
    from openstereo_amd.utils.weights import synth_state_dict, synth_images
    fake = sys.modules.setdefault("timm", types.ModuleType("timm"))
    fake.create_model = FP.create_model
    img, _ = synth_images(2, 128, 256, seed=37)
    out = {}
    for tag, modname, clsname, seed in (("stereobase", "stereo.modeling.models.stereobase.backbone", "Feature", 61),
                                        ("igev", "stereo.modeling.models.igev.extractor", "Feature", 62),
                                        ("lightstereo", "stereo.modeling.models.lightstereo.backbone", "Backbone", 63)):
        mod = importlib.import_module(modname)
        mod.timm = fake                                   # (the module did `import timm` at import time)
        net = getattr(mod, clsname)().eval()
        net.load_state_dict(synth_state_dict(net, seed=seed, gain=0.9))
        outs = net(img)
        for i, t in enumerate(outs):
            out[f"{tag}_out{i}"] = t
        out[f"{tag}_keys"] = np.array(sorted(net.state_dict().keys()))
        print(f"{tag} pyramid: {[tuple(t.shape) for t in outs]}, std {[round(float(t.std()), 3) for t in outs]}, {len(net.state_dict())} keys")
    save("feature_pyramid.npz", **out)


def gen_e2e_dormant():
    """StereoBase with the dormant volume switches (stereobase_gru.py:22-23,152-159; no shipped config sets them): the reference's own class,
    (a) USE_SUB_VOLUME + USE_INTERLACED_VOLUME on top of gwc + concat (8 + 16 + 1 + 8 = 33 volume channels), (b) gwc + interlaced only
    (16 channels: the fused NDHWC route).  build_sub_volume's device='cuda' zeros are redirected as in gen_dormant.  Also the
    InterlacedVolume module alone (cost_volume.py:120-169).  96x192, MAX_DISP 32, 3 iterations."""
    from openstereo_amd.utils.weights import synth_state_dict, synth_images
    from openstereo_amd.models.stereo_models import StubFeature
    from stereo.modeling.cost_volume import cost_volume as cv

    class _CpuTorch:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def zeros(*a, **k):
            k.pop("device", None)
            return torch.zeros(*a, **k)
    feat = lambda *a, **k: StubFeature((48, 64, 192, 160))
    L, R = synth_images(1, 96, 192, seed=35, max_shift=6.0)
    out = {}
    iv = cv.InterlacedVolume(8).eval()
    iv.load_state_dict(synth_state_dict(iv, seed=23, gain=0.9))
    fl, fr = rnd((2, 96, 7, 19), 311), rnd((2, 96, 7, 19), 312)
    out["interlaced_alone"] = iv(fl, fr, 6)
    print("InterlacedVolume alone: max", out["interlaced_alone"].abs().max().item())
    base = Cfg(MAX_DISP=32, NUM_GROUPS=8, USE_GWC_VOLUME=True, CONCAT_CHANNELS=8, INTERLACED_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128],
               N_DOWNSAMPLE=2, N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=False, TRAIN_ITERS=3, EVAL_ITERS=3)
    real = cv.torch
    cv.torch = _CpuTorch()
    try:
        for tag, flags in (("all", dict(USE_CONCAT_VOLUME=True, USE_SUB_VOLUME=True, USE_INTERLACED_VOLUME=True)),
                           ("inter", dict(USE_CONCAT_VOLUME=False, USE_SUB_VOLUME=False, USE_INTERLACED_VOLUME=True))):
            net = _reference_model("stereo.modeling.models.stereobase.stereobase_gru", "StereoBase", Cfg(base, **flags), {"Feature": feat})
            net.load_state_dict(synth_state_dict(net, seed=53, head_gain=20.0, gain=0.9))
            r = net({"left": L, "right": R})
            out[f"sb_{tag}_disp"], out[f"sb_{tag}_init"] = r["disp_pred"], r["init_disp"]
            print(f"StereoBase dormant [{tag}]: disp range", r["disp_pred"].min().item(), r["disp_pred"].max().item(), "std", r["disp_pred"].std().item())
    finally:
        cv.torch = real
    save("e2e_dormant.npz", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="", help="regenerate a single fixture group: lightstereo | igev_update | at_size | preprocess | dormant | e2e | e2e_dormant | e2e_train_at_size | feature_pyramid | unit_gain | context_encoder")
    args = ap.parse_args()
    import_reference()
    torch.set_grad_enabled(False)
    from openstereo_amd.utils.weights import synth_state_dict, synth_images
    if args.only == "lightstereo":
        gen_lightstereo()
        return
    if args.only == "igev_update":
        gen_igev_update()
        return
    if args.only == "at_size":
        gen_at_size()
        return
    if args.only == "preprocess":
        gen_preprocess()
        return
    if args.only == "dormant":
        gen_dormant()
        return
    if args.only == "e2e":
        gen_e2e()
        return
    if args.only == "e2e_dormant":
        gen_e2e_dormant()
        return
    if args.only == "feature_pyramid":
        gen_feature_pyramid()
        return
    if args.only == "e2e_train_at_size":
        gen_e2e_train_at_size()
        return
    if args.only == "unit_gain":
        gen_unit_gain()
        return
    if args.only == "context_encoder":
        gen_context_encoder()
        return

    # ------------------------------------------------------------------ volumes (a1-a4)
    from stereo.modeling.cost_volume import cost_volume as cv
    from stereo.modeling.models.gwcnet.gwcnet_cost_processor import GwcVolumeCostProcessor
    from stereo.modeling.models.igev import submodule as igev_sub
    from stereo.modeling.models.psmnet.psmnet_cost_processor import cat_fms

    out = {}
    for tag, (B, C, H, W, D, G) in {"a": (2, 16, 5, 23, 9, 4), "narrow": (1, 24, 3, 6, 9, 2),
                                    "k12": (1, 24, 4, 19, 8, 2)}.items():
        L, R = rnd((B, C, H, W), 100 + len(tag)), rnd((B, C, H, W), 200 + len(tag))
        out[f"{tag}_meta"] = np.array([B, C, H, W, D, G])
        out[f"{tag}_L"], out[f"{tag}_R"] = L, R
        out[f"{tag}_gwc"] = cv.build_gwc_volume(L, R, D, G)
        out[f"{tag}_concat"] = cv.build_concat_volume(L, R, D)
        out[f"{tag}_corr"] = cv.correlation_volume(L, R, D)
        out[f"{tag}_corr2"] = cv.build_corr_volume(L, R, D)
        out[f"{tag}_igev_gwc"] = igev_sub.build_gwc_volume(L, R, D, G)
        out[f"{tag}_igev_concat"] = igev_sub.build_concat_volume(L, R, D)
        out[f"{tag}_psm_cat"] = cat_fms(L, R, max_disp=D, start_disp=0, dilation=1)
        cp = GwcVolumeCostProcessor(maxdisp=D * 4, downsample=4, num_groups=G, use_concat_volume=True)
        feats = {"ref_feature": {"gwc_feature": L, "concat_feature": L[:, :6]},
                 "tgt_feature": {"gwc_feature": R, "concat_feature": R[:, :6]}}
        out[f"{tag}_gwcnet_volume"] = cp(feats)["cost_volume"]
    save("volumes.npz", **out)

    # ------------------------------------------------------------------ regression (a10-a12)
    import torch.nn.functional as F
    from stereo.modeling.disp_pred.disp_regression import disparity_regression as dr_keep
    from stereo.modeling.models.gwcnet.gwcnet_disp_processor import disparity_regression as dr_nokeep
    from stereo.modeling.models.psmnet.psmnet_disp_processor import FasterSoftArgmin
    cost = rnd((2, 12, 7, 9), 7) * 3.0
    prob = F.softmax(cost, dim=1)
    low = rnd((2, 1, 6, 5, 7), 8) * 2.0
    up_f = F.interpolate(low, [24, 20, 28], mode="trilinear")
    up_t = F.interpolate(low, [24, 20, 28], mode="trilinear", align_corners=True)
    low2 = rnd((1, 1, 5, 4, 6), 9) * 2.0           # non-integer scale factors
    up_odd = F.interpolate(low2, [17, 13, 21], mode="trilinear")
    save("regression.npz", cost=cost, prob=prob,
         reg_keep=dr_keep(prob, 12), reg_nokeep=dr_nokeep(prob, 12),
         faster_softargmin=FasterSoftArgmin(max_disp=12, start_disp=0, dilation=1, alpha=1.0, normalize=True)(cost),
         low=low,
         up_false=dr_nokeep(F.softmax(up_f.squeeze(1), dim=1), 24),
         up_true=dr_nokeep(F.softmax(up_t.squeeze(1), dim=1), 24),
         low2=low2, up_odd=dr_nokeep(F.softmax(up_odd.squeeze(1), dim=1), 17))

    # ------------------------------------------------------------------ geometry-encoding volume (a5)
    from stereo.modeling.models.stereobase.gru_blocks import CombinedGeoEncodingVolume
    from stereo.modeling.models.igev.geometry import Combined_Geo_Encoding_Volume
    f1, f2 = rnd((2, 12, 5, 14), 41), rnd((2, 12, 5, 14), 42)
    gvol = rnd((2, 6, 10, 5, 14), 43)
    dsp = (rnd((2, 1, 5, 14), 44).abs() * 3.0)
    crd = torch.arange(14, dtype=torch.float32).view(1, 1, 14, 1).repeat(2, 5, 1, 1)
    gev = CombinedGeoEncodingVolume(f1, f2, gvol, num_levels=2, radius=4)
    lk = gev(dsp, crd)
    assert torch.equal(lk, Combined_Geo_Encoding_Volume(f1, f2, gvol, num_levels=2, radius=4)(dsp, crd))
    save("geo_encoding.npz", f1=f1, f2=f2, geo=gvol, disp=dsp, coords=crd, lookup=lk,
         corr=CombinedGeoEncodingVolume.corr(f1, f2), lookup2=gev(dsp * 2.5 + 1.0, crd))

    # ------------------------------------------------------------------ context_upsample (a13)
    from stereo.modeling.disp_refinement.disp_refinement import context_upsample as cu_shared
    from stereo.modeling.models.stereobase.igev_blocks import context_upsample as cu_sb
    dl, lg = rnd((2, 1, 6, 9), 31).abs() * 10, rnd((2, 9, 24, 36), 32)
    wts = F.softmax(lg, 1)
    assert torch.equal(cu_shared(dl * 4., wts), cu_sb(dl * 4., wts))
    save("context_upsample.npz", disp_low=dl, logits=lg, weights=wts, out=cu_shared(dl * 4., wts),
         out_s2=cu_shared(dl[:, :, :3, :4], wts[:, :, :6, :8], scale_factor=2))

    # ------------------------------------------------------------------ GwcNet hourglass + disp processor (a6, a10)
    from stereo.modeling.models.gwcnet.hourglass import Hourglass
    from stereo.modeling.models.gwcnet.gwcnet_disp_processor import GwcDispProcessor
    hg = Hourglass(8).eval()
    hg.load_state_dict(synth_state_dict(hg, seed=3))
    x = rnd((1, 8, 8, 8, 16), 11)
    save("gwc_hourglass.npz", x=x, y=hg(x))

    dp = GwcDispProcessor(maxdisp=32, downsample=4, num_groups=40, use_concat_volume=True, concat_channels=12).eval()
    dp.load_state_dict(synth_state_dict(dp, seed=4))
    taps = {}
    dp.dres2.register_forward_pre_hook(lambda m, i: taps.__setitem__("cost0", i[0].clone()))
    dp.dres2.register_forward_hook(lambda m, i, o: taps.__setitem__("out1", o.clone()))
    dp.dres4.register_forward_hook(lambda m, i, o: taps.__setitem__("out3", o.clone()))
    dp.classif3.register_forward_hook(lambda m, i, o: taps.__setitem__("cost3", o.clone()))
    vol = rnd((1, 64, 8, 8, 16), 12).abs()
    disp = dp({"cost_volume": vol, "left": torch.zeros(1, 3, 32, 64)})["inference_disp"]["disp_est"]
    save("gwc_disp.npz", volume=vol, disp=disp, **taps)

    # ------------------------------------------------------------------ full GwcNet, small image
    from stereo.modeling.models.gwcnet.gwcnet import GwcNet
    cfg = Cfg(MAX_DISP=192, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=12, DOWNSAMPLE=4, NUM_GROUPS=40)
    net = GwcNet(cfg).eval()
    net.load_state_dict(synth_state_dict(net, seed=0))
    taps = {}
    net.Backbone.register_forward_hook(lambda m, i, o: taps.update(
        left_gwc=o["ref_feature"]["gwc_feature"].clone(), right_gwc=o["tgt_feature"]["gwc_feature"].clone(),
        left_cat=o["ref_feature"]["concat_feature"].clone(), right_cat=o["tgt_feature"]["concat_feature"].clone()))
    net.DispProcessor.classif3.register_forward_hook(lambda m, i, o: taps.__setitem__("cost3", o.clone()))
    L, R = synth_images(1, 64, 128, seed=1)
    disp = net({"left": L, "right": R})["disp_pred"]
    print("small GwcNet disp range", disp.min().item(), disp.max().item(), disp.std().item())
    save("gwcnet_small.npz", disp=disp, **taps)

    # ------------------------------------------------------------------ StereoBase / IGEV hourglass (a8)
    from stereo.modeling.models.stereobase.hourglass import Hourglass as SBHourglass
    sys.modules.setdefault("timm", types.ModuleType("timm"))        # igev/extractor.py imports timm at module level only
    from stereo.modeling.models.igev.igev_stereo import hourglass as IGEVHourglass
    feats = [None, rnd((1, 64, 8, 16), 21), rnd((1, 192, 4, 8), 22), None]
    sbh = SBHourglass(24, [96, 64, 192, 120]).eval()
    sbh.load_state_dict(synth_state_dict(sbh, seed=6))
    xs = rnd((1, 24, 8, 16, 32), 23)
    fs = feats[:3] + [rnd((1, 120, 2, 4), 24)]
    multi = sbh(xs, fs, return_multi=True)
    save("stereobase_hourglass.npz", x=xs, f1=fs[1], f2=fs[2], f3=fs[3], y=multi[0], y1=multi[1], y2=multi[2])
    igh = IGEVHourglass(8).eval()
    igh.load_state_dict(synth_state_dict(igh, seed=7))
    xi = rnd((1, 8, 8, 16, 32), 25)
    fi = feats[:3] + [rnd((1, 160, 2, 4), 26)]
    save("igev_hourglass.npz", x=xi, f1=fi[1], f2=fi[2], f3=fi[3], y=igh(xi, fi))

    # ------------------------------------------------------------------ LightStereo 2-D aggregation (a9)
    gen_lightstereo()
    gen_igev_update()
    gen_at_size()
    gen_preprocess()
    gen_dormant()
    gen_e2e()
    gen_unit_gain()
    gen_context_encoder()
    gen_e2e_dormant()

    # ------------------------------------------------------------------ PSMNet, BASELINE configs[0]: 256x512, D=64
    from stereo.modeling.models.psmnet.psmnet import PSMNet
    psm = PSMNet(Cfg(MAX_DISP=64)).eval()
    psm.load_state_dict(synth_state_dict(psm, seed=0, head_gain=3.0), strict=False)   # keeps the frozen linspace kernel
    ptaps = {}
    psm.Backbone.register_forward_hook(lambda m, i, o: ptaps.update(
        left_feature=o["ref_feature"].clone(), right_feature=o["tgt_feature"].clone()))
    psm.CostProcessor.aggregator.dres4.register_forward_hook(      # subsampled tap of the last hourglass output
        lambda m, i, o: ptaps.__setitem__("hg3_out_sub", o[0][:, :, ::2, ::4, ::4].clone()))
    L, R = synth_images(1, 256, 512, seed=1, max_shift=16.0)
    t = time.time()
    out = psm({"left": L, "right": R})
    print(f"PSMNet 256x512 D=64 reference forward: {time.time() - t:.1f} s; disp3 range "
          f"{out['disp_pred'].min().item():.2f}..{out['disp_pred'].max().item():.2f} std {out['disp_pred'].std().item():.2f}")
    save("psmnet_256x512.npz", disp1=out["train_preds"][0], disp2=out["train_preds"][1], disp3=out["train_preds"][2],
         **ptaps)

    if args.full:
        L, R = synth_images(1, 544, 960, seed=1)
        t = time.time()
        taps.clear()
        disp = net({"left": L, "right": R})["disp_pred"]
        print(f"full GwcNet reference forward: {time.time() - t:.1f} s on {torch.get_num_threads()} threads;"
              f" disp range {disp.min().item():.2f}..{disp.max().item():.2f} std {disp.std().item():.2f}")
        save("gwcnet_full_disp.npz", disp=disp.numpy().astype(np.float32),
             cost3=taps["cost3"].numpy().astype(np.float32)[:, :, ::4, ::4, ::4])


if __name__ == "__main__":
    main()
