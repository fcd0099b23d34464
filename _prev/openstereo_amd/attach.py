"""Attach the engine to an unmodified OpenStereo checkout (SURVEY 8a patch list).

    import openstereo_amd.attach as A
    A.patch_reference()            # rebinds every per-model copy of the hot-path helper FUNCTIONS
    A.patch_reference_modules()    # grafts the engine forwards onto the reference's own nn.Module CLASSES
    ...build / load reference models as usual: `GwcNet(cfg)`, `PSMNet(cfg)`, ... run on the engine...
    A.unpatch_reference()

Nothing in the reference tree is edited; only module / class attributes are rebound.  Works with whatever
subset of reference modules is importable (models that need timm etc. are skipped).

Functions: every replacement is DIFFERENTIABLE -- when autograd is recording and an argument requires grad the
call goes through `openstereo_amd.autograd` (forward and backward on the engine's kernels), otherwise through the
non-recording `openstereo_amd.ops` entry.  A reference trainer running under the patch therefore trains exactly
as before (cost_volume.py:68-92 and disp_regression.py:8-12 are differentiable compositions).

Classes: the engine mirrors in `openstereo_amd.models` use the reference's attribute names, so their methods can
be grafted onto the reference classes without rebuilding a model: parameters stay where they are (same
Parameter objects -> optimisers, DDP, checkpoints unaffected), packed weights are cached per module and
rebuilt when a parameter or BatchNorm statistic changes (engine.cached_pack).  In training mode the grafted
forwards take the autograd path (BatchNorm batch statistics, ReLU etc. as torch modules); modules without a
training path keep the reference's original forward for training.
"""
from __future__ import annotations

import importlib
import sys
import types

import torch

from . import ops

_saved = []
_MISSING = object()


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


# ----------------------------------------------------------------------------- differentiable drop-in functions
def build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """cost_volume.py:68-78"""
    if _needs_grad(refimg_fea, targetimg_fea):
        from . import autograd as AG
        return AG.build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups).to(refimg_fea.dtype)
    return ops.build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups)


def build_concat_volume(refimg_fea, targetimg_fea, maxdisp, mask_left=True):
    """cost_volume.py:81-92 (mask_left=False: igev/submodule.py:216-227)"""
    if _needs_grad(refimg_fea, targetimg_fea):
        from . import autograd as AG
        return AG.build_concat_volume(refimg_fea, targetimg_fea, maxdisp, mask_left).to(refimg_fea.dtype)
    return ops.build_concat_volume(refimg_fea, targetimg_fea, maxdisp, mask_left=mask_left)


def correlation_volume(left_feature, right_feature, max_disp):
    """cost_volume.py:32-41"""
    if _needs_grad(left_feature, right_feature):
        from . import autograd as AG
        return AG.correlation_volume(left_feature, right_feature, max_disp).to(left_feature.dtype)
    return ops.correlation_volume(left_feature, right_feature, max_disp)


def build_corr_volume(img_left, img_right, max_disp):
    """cost_volume.py:95-105 (planes d >= W repeat plane 0, see ops.build_corr_volume)"""
    if _needs_grad(img_left, img_right):
        vol = correlation_volume(img_left, img_right, max_disp)
        W = img_left.shape[-1]
        return vol if max_disp <= W else torch.cat((vol[:, :W], vol[:, :1].expand(-1, max_disp - W, -1, -1)), 1)
    return ops.build_corr_volume(img_left, img_right, max_disp)


def cat_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1):
    """psmnet_cost_processor.py:9-50 (always fp32, like the reference's buffer)"""
    if _needs_grad(reference_fm, target_fm):
        if start_disp != 0 or dilation != 1:             # no backward kernel for the general sampling: the reference's own (differentiable) code
            for mod, attr, old in _saved:
                if attr == "cat_fms" and callable(old):
                    return old(reference_fm, target_fm, max_disp, start_disp, dilation)
            raise NotImplementedError("cat_fms with start_disp / dilation has no autograd path on the engine")
        from . import autograd as AG
        return AG.build_concat_volume(reference_fm.float(), target_fm.float(), max_disp)
    return ops.cat_fms(reference_fm, target_fm, max_disp, start_disp, dilation)


def disparity_regression(x, maxdisp, keepdim=True):
    """disp_regression.py:8-12 (keepdim=True) / gwcnet_disp_processor.py:22-26 (keepdim=False)"""
    if _needs_grad(x):
        from . import autograd as AG
        assert len(x.shape) == 4
        return AG.disparity_regression(x, maxdisp, keepdim).to(x.dtype)
    return ops.disparity_regression(x, maxdisp, keepdim)


def context_upsample(disp_low, up_weights, scale_factor=4):
    """disp_refinement.py:194-204.  Inference: fused engine kernel; when gradients are required: the same arithmetic as
    a torch composition (unfold -> nearest -> weighted sum), which autograd differentiates (HBM-bound elementwise ops)."""
    if _needs_grad(disp_low, up_weights):
        import torch.nn.functional as F
        b, c, h, w = disp_low.shape
        u = F.unfold(disp_low.reshape(b, c, h, w), 3, 1, 1).reshape(b, -1, h, w)
        u = F.interpolate(u, (h * scale_factor, w * scale_factor), mode="nearest").reshape(b, 9, h * scale_factor, w * scale_factor)
        return (u * up_weights).sum(1)
    return ops.context_upsample(disp_low, up_weights, scale_factor)


def _dormant(name):
    """Engine version of a dormant volume helper for inference; with gradients required it defers to the function it replaced."""
    def fn(*a, **k):
        if _needs_grad(*[t for t in a if isinstance(t, torch.Tensor)]):
            for mod, attr, old in _saved:
                if attr == name and callable(old) and mod.__name__.endswith("cost_volume.cost_volume"):
                    return old(*a, **k)
            raise NotImplementedError(f"openstereo_amd: {name} has no autograd path on the engine")
        return getattr(ops, name)(*a, **k)
    fn.__name__ = name
    return fn


# (module, attribute, replacement)
def _targets():
    igev_concat = lambda l, r, d: build_concat_volume(l, r, d, mask_left=False)   # igev/submodule.py:216-227
    reg_keep = lambda x, maxdisp: disparity_regression(x, maxdisp, keepdim=True)
    reg_nokeep = lambda x, maxdisp: disparity_regression(x, maxdisp, keepdim=False)
    cv = "stereo.modeling.cost_volume.cost_volume"
    return [
        (cv, "build_gwc_volume", build_gwc_volume), (cv, "build_concat_volume", build_concat_volume),
        (cv, "correlation_volume", correlation_volume), (cv, "build_corr_volume", build_corr_volume),
        # dormant variants (inference; no autograd path): engine versions fall back to the reference's own code when gradients are needed
        (cv, "compute_volume", _dormant("compute_volume")), (cv, "build_sub_volume", _dormant("build_sub_volume")),
        ("stereo.modeling.models.stereobase.stereobase_gru", "build_sub_volume", _dormant("build_sub_volume")),
        ("stereo.modeling.disp_pred.disp_regression", "disparity_regression", reg_keep),
        ("stereo.modeling.disp_refinement.disp_refinement", "context_upsample", context_upsample),
        # names already imported into model namespaces
        ("stereo.modeling.models.stereobase.stereobase_gru", "build_gwc_volume", build_gwc_volume),
        ("stereo.modeling.models.stereobase.stereobase_gru", "build_concat_volume", build_concat_volume),
        ("stereo.modeling.models.stereobase.stereobase_gru", "disparity_regression", reg_keep),
        ("stereo.modeling.models.stereobase.stereobase_gru", "context_upsample", context_upsample),
        ("stereo.modeling.models.stereobase.igev_blocks", "context_upsample", context_upsample),
        ("stereo.modeling.models.lightstereo.lightstereo", "correlation_volume", correlation_volume),
        ("stereo.modeling.models.lightstereo.lightstereo", "disparity_regression", reg_keep),
        ("stereo.modeling.models.lightstereo.lightstereo", "context_upsample", context_upsample),
        ("stereo.modeling.models.gwcnet.gwcnet_disp_processor", "disparity_regression", reg_nokeep),
        ("stereo.modeling.models.psmnet.psmnet_cost_processor", "cat_fms", cat_fms),
        ("stereo.modeling.models.psmnet.psmnet_disp_processor", "FasterSoftArgmin", ops.FasterSoftArgmin),
        ("stereo.modeling.models.igev.submodule", "build_gwc_volume", build_gwc_volume),
        ("stereo.modeling.models.igev.submodule", "build_concat_volume", igev_concat),
        ("stereo.modeling.models.igev.submodule", "disparity_regression", reg_keep),
        ("stereo.modeling.models.igev.submodule", "context_upsample", context_upsample),
        ("stereo.modeling.models.igev.igev_stereo", "build_gwc_volume", build_gwc_volume),
        ("stereo.modeling.models.igev.igev_stereo", "build_concat_volume", igev_concat),
        ("stereo.modeling.models.igev.igev_stereo", "disparity_regression", reg_keep),
        ("stereo.modeling.models.igev.igev_stereo", "context_upsample", context_upsample),
    ]


def stub_reference_packages(ref_root: str):
    """Register stub parent packages so `stereo.modeling.*` sub-modules import without executing
    stereo/modeling/__init__.py (which needs cv2/timm/easydict ...)."""
    import os
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    for name, path in [("stereo", "stereo"), ("stereo.modeling", "stereo/modeling"),
                       ("stereo.modeling.models", "stereo/modeling/models")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(ref_root, path)]
            sys.modules[name] = m


def patch_reference(verbose: bool = False) -> list[str]:
    """Rebind the hot-path helpers in every importable reference module. Returns what was patched."""
    done = []
    for mod_name, attr, repl in _targets():
        try:
            mod = importlib.import_module(mod_name)
        except Exception as ex:                       # model family not importable here (timm, cv2, ...)
            if verbose:
                print(f"[attach] skip {mod_name}: {type(ex).__name__}: {ex}")
            continue
        if not hasattr(mod, attr):
            continue
        _saved.append((mod, attr, getattr(mod, attr)))
        setattr(mod, attr, repl)
        done.append(f"{mod_name}.{attr}")
    # GwcNet keeps its constructors as *methods*; PSMCostProcessor binds cat_fms into a functools.partial at
    # construction (psmnet_cost_processor.py:227), so instances built after the patch pick up the replacement
    try:
        cp = importlib.import_module("stereo.modeling.models.gwcnet.gwcnet_cost_processor").GwcVolumeCostProcessor
        for name, fn in (("build_gwc_volume", lambda self, l, r: build_gwc_volume(l, r, self.maxdisp // self.downsample, self.num_groups)),
                         ("build_concat_volume", lambda self, l, r: build_concat_volume(l, r, self.maxdisp // self.downsample))):
            _saved.append((cp, name, getattr(cp, name)))
            setattr(cp, name, fn)
            done.append(f"GwcVolumeCostProcessor.{name}")
    except Exception:
        pass
    return done


def _graft(ref_cls, eng_cls, names, train_fallback=False):
    """Give a reference nn.Module class the engine methods of its mirror (same attribute layout).
    train_fallback: the mirror has no training path -- keep the reference's own forward for training mode / when
    gradients are required, instead of raising."""
    orig_forward = ref_cls.__dict__.get("forward")
    for n in names:
        if n in eng_cls.__dict__ or hasattr(eng_cls, n):
            _saved.append((ref_cls, n, ref_cls.__dict__.get(n, _MISSING)))
            setattr(ref_cls, n, eng_cls.__dict__[n] if n in eng_cls.__dict__ else getattr(eng_cls, n))
    if train_fallback and orig_forward is not None and "forward" in names:
        eng_forward = eng_cls.__dict__["forward"]

        def forward(self, *a, **k):
            if self.training or _needs_grad(*a, *(p for p in self.parameters())):
                return orig_forward(self, *a, **k)
            return eng_forward(self, *a, **k)
        setattr(ref_cls, "forward", forward)
    for n in ("_eng", "_mask", "_packed", "_pk", "_cls"):
        if n not in ref_cls.__dict__:
            _saved.append((ref_cls, n, _MISSING))
            setattr(ref_cls, n, None)


M = "stereo.modeling.models."


def _graft_plan():
    from .models import gwcnet as GW, psmnet as PSM, igev_style as IG, lightstereo as LS, igev_update as UP
    fwd = ("forward", "forward_cl", "forward_train", "_pack", "reset_engine")
    gru = ("_packs", "new_level", "step", "_levels")          # per-level state buffers of the update block (igev_update.py)
    return [
        # (reference module, {reference class: (mirror class, methods, keep the reference forward for training)})
        (M + "gwcnet.hourglass", {"Hourglass": (GW.Hourglass, fwd, False)}),                                   # hourglass.py:46-56
        (M + "gwcnet.gwcnet_disp_processor", {"GwcDispProcessor": (GW.GwcDispProcessor, fwd + ("aggregate_cl",), False)}),   # :83-140
        (M + "gwcnet.gwcnet_cost_processor", {"GwcVolumeCostProcessor": (GW.GwcVolumeCostProcessor, ("forward",), False)}),  # :55-68
        (M + "gwcnet.gwcnet_backbone", {"feature_extraction": (GW._Features, ("forward_cl", "_pack"), False),
                                        "GwcNet": (GW.GwcBackbone, ("forward", "forward_cl", "use_engine"), False)}),  # :78-112
        (M + "gwcnet.gwcnet", {"GwcNet": (GW.GwcNet, ("forward", "reset_engine"), False)}),                     # gwcnet.py:27-39
        (M + "psmnet.psmnet_cost_processor", {"Hourglass": (PSM.Hourglass, fwd, False),                         # :108-132
                                              "PSMAggregator": (PSM.PSMAggregator, fwd + ("aggregate_cl", "aggregate_train"), False),   # :182-221
                                              "PSMCostProcessor": (PSM.PSMCostProcessor, ("forward",), False)}),
        (M + "psmnet.psmnet_disp_processor", {"PSMDispProcessor": (PSM.PSMDispProcessor, ("forward",), False)}),
        (M + "psmnet.psmnet_backbone", {"PSMNet": (PSM.PSMBackbone, ("forward", "forward_cl", "_pack", "reset_engine", "use_engine"), False)}),
        (M + "stereobase.igev_blocks", {"FeatureAtt": (IG.FeatureAtt, ("logits",), False)}),
        (M + "stereobase.hourglass", {"Hourglass": (IG.Hourglass, fwd + ("gate_logits", "_unit_train"), False)}),   # hourglass.py:79-104
        (M + "igev.submodule", {"FeatureAtt": (IG.IGEVFeatureAtt, ("logits",), False)}),
        (M + "igev.igev_stereo", {"hourglass": (IG.hourglass, ("forward", "forward_cl", "forward_train", "_unit_train", "_packed_layers", "reset_engine"), False)}),  # :51-76
        (M + "lightstereo.aggregation", {c: (getattr(LS, c), fwd, False) for c in ("Aggregation", "MobileV2Residual", "AttentionModule")}),
        (M + "igev.update", {c: (getattr(UP, c), fwd + gru, False) for c in ("ConvGRU", "BasicMotionEncoder", "DispHead", "BasicMultiUpdateBlock")}),
        (M + "stereobase.gru_blocks", {c: (getattr(UP, c), fwd + gru, False) for c in ("ConvGRU", "BasicMotionEncoder", "DispHead", "BasicMultiUpdateBlock")}),
    ]


def patch_reference_modules(verbose: bool = False) -> list[str]:
    """Graft the engine forwards onto the reference's OWN module classes (no model rebuild, parameters stay where
    they are): the 3-D aggregation classes of GwcNet / PSMNet / StereoBase / IGEV (SURVEY 8a patch list), GwcNet's and
    PSMNet's stage containers and 2-D backbones, LightStereo's `Aggregation` and the IGEV / StereoBase update block.
    Works because the mirrors use the reference's attribute names."""
    done = []
    for mod_name, classes in _graft_plan():
        try:
            mod = importlib.import_module(mod_name)
        except Exception as ex:
            if verbose:
                print(f"[attach] skip {mod_name}: {type(ex).__name__}: {ex}")
            continue
        for cname, (eng_cls, names, train_fallback) in classes.items():
            if hasattr(mod, cname):
                _graft(getattr(mod, cname), eng_cls, names, train_fallback)
                done.append(f"{mod_name}.{cname}")
    return done


def unpatch_reference():
    while _saved:
        obj, attr, old = _saved.pop()
        if old is _MISSING:
            if attr in getattr(obj, "__dict__", {}):
                delattr(obj, attr)
        else:
            setattr(obj, attr, old)


def attach_gwcnet(model):
    """Build an engine GwcNet that SHARES the reference model's Parameter and buffer objects (the module trees have
    identical names, so every tensor of the engine model is rebound to the reference's): training either one updates
    both, and there is no second copy of the weights in HBM.  Returns the engine model in the reference's mode."""
    from .models.gwcnet import GwcNet, _Cfg
    cp, dp = model.CostProcessor, model.DispProcessor
    cfg = _Cfg(MAX_DISP=model.maxdisp, USE_CONCAT_VOLUME=cp.use_concat_volume, CONCAT_CHANNELS=dp.concat_channels,
               DOWNSAMPLE=cp.downsample, NUM_GROUPS=cp.num_groups)
    eng = GwcNet(cfg)
    ref_mods = dict(model.named_modules())
    for name, m in eng.named_modules():
        r = ref_mods.get(name)
        if r is None:
            continue
        for k in list(m._parameters):
            if k in r._parameters:
                m._parameters[k] = r._parameters[k]
        for k in list(m._buffers):
            if k in r._buffers:
                m._buffers[k] = r._buffers[k]
    missing = set(eng.state_dict()) ^ set(model.state_dict())
    assert not missing, f"state_dict layouts differ: {sorted(missing)[:5]}"
    return eng.train(model.training)
