"""Attach the engine to an unmodified OpenStereo checkout (SURVEY 8a patch list).

    import openstereo_amd.attach as A
    A.patch_reference()            # rebinds every per-model copy of the hot-path helpers
    ...build / load reference models as usual...
    A.patch_reference_modules()    # optional: engine forwards grafted onto the reference's LightStereo / GRU module classes
    A.attach_gwcnet(model)         # optional: swap CostProcessor / DispProcessor / Backbone forwards
    A.unpatch_reference()

Nothing in the reference tree is edited; only module attributes are rebound.  Works with whatever
subset of reference modules is importable (models that need timm etc. are skipped).
"""
from __future__ import annotations

import importlib
import sys
import types

from . import ops

_saved = []

# (module, attribute, replacement)
def _targets():
    igev_concat = lambda l, r, d: ops.build_concat_volume(l, r, d, mask_left=False)   # igev/submodule.py:216-227
    reg_keep = lambda x, maxdisp: ops.disparity_regression(x, maxdisp, keepdim=True)
    reg_nokeep = lambda x, maxdisp: ops.disparity_regression(x, maxdisp, keepdim=False)
    cv = "stereo.modeling.cost_volume.cost_volume"
    return [
        (cv, "build_gwc_volume", ops.build_gwc_volume), (cv, "build_concat_volume", ops.build_concat_volume),
        (cv, "correlation_volume", ops.correlation_volume), (cv, "build_corr_volume", ops.build_corr_volume),
        ("stereo.modeling.disp_pred.disp_regression", "disparity_regression", reg_keep),
        ("stereo.modeling.disp_refinement.disp_refinement", "context_upsample", ops.context_upsample),
        # names already imported into model namespaces
        ("stereo.modeling.models.stereobase.stereobase_gru", "build_gwc_volume", ops.build_gwc_volume),
        ("stereo.modeling.models.stereobase.stereobase_gru", "build_concat_volume", ops.build_concat_volume),
        ("stereo.modeling.models.stereobase.stereobase_gru", "disparity_regression", reg_keep),
        ("stereo.modeling.models.stereobase.stereobase_gru", "context_upsample", ops.context_upsample),
        ("stereo.modeling.models.stereobase.igev_blocks", "context_upsample", ops.context_upsample),
        ("stereo.modeling.models.lightstereo.lightstereo", "correlation_volume", ops.correlation_volume),
        ("stereo.modeling.models.lightstereo.lightstereo", "disparity_regression", reg_keep),
        ("stereo.modeling.models.lightstereo.lightstereo", "context_upsample", ops.context_upsample),
        ("stereo.modeling.models.gwcnet.gwcnet_disp_processor", "disparity_regression", reg_nokeep),
        ("stereo.modeling.models.psmnet.psmnet_cost_processor", "cat_fms", ops.cat_fms),
        ("stereo.modeling.models.psmnet.psmnet_disp_processor", "FasterSoftArgmin", ops.FasterSoftArgmin),
        ("stereo.modeling.models.igev.submodule", "build_gwc_volume", ops.build_gwc_volume),
        ("stereo.modeling.models.igev.submodule", "build_concat_volume", igev_concat),
        ("stereo.modeling.models.igev.submodule", "disparity_regression", reg_keep),
        ("stereo.modeling.models.igev.submodule", "context_upsample", ops.context_upsample),
        ("stereo.modeling.models.igev.igev_stereo", "build_gwc_volume", ops.build_gwc_volume),
        ("stereo.modeling.models.igev.igev_stereo", "build_concat_volume", igev_concat),
        ("stereo.modeling.models.igev.igev_stereo", "disparity_regression", reg_keep),
        ("stereo.modeling.models.igev.igev_stereo", "context_upsample", ops.context_upsample),
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
    # GwcNet keeps its constructors as *methods*
    try:
        cp = importlib.import_module("stereo.modeling.models.gwcnet.gwcnet_cost_processor").GwcVolumeCostProcessor
        for name, fn in (("build_gwc_volume", lambda self, l, r: ops.build_gwc_volume(l, r, self.maxdisp // self.downsample, self.num_groups)),
                         ("build_concat_volume", lambda self, l, r: ops.build_concat_volume(l, r, self.maxdisp // self.downsample))):
            _saved.append((cp, name, getattr(cp, name)))
            setattr(cp, name, fn)
            done.append(f"GwcVolumeCostProcessor.{name}")
    except Exception:
        pass
    return done


def _graft(ref_cls, eng_cls, names):
    """Give a reference nn.Module class the engine methods of its mirror (same attribute layout)."""
    for n in names:
        if hasattr(eng_cls, n):
            _saved.append((ref_cls, n, ref_cls.__dict__.get(n, _MISSING)))
            setattr(ref_cls, n, eng_cls.__dict__[n] if n in eng_cls.__dict__ else getattr(eng_cls, n))
    for n in ("_eng", "_mask"):
        if n not in ref_cls.__dict__:
            _saved.append((ref_cls, n, _MISSING))
            setattr(ref_cls, n, None)


def patch_reference_modules(verbose: bool = False) -> list[str]:
    """Graft the engine forwards onto the reference's OWN module classes (no model rebuild, parameters
    stay where they are): LightStereo `Aggregation` / `MobileV2Residual` / `AttentionModule`
    (models/lightstereo/aggregation.py) and the IGEV / StereoBase update block (`ConvGRU`,
    `BasicMotionEncoder`, `DispHead`, `BasicMultiUpdateBlock`; models/igev/update.py,
    models/stereobase/gru_blocks.py).  Works because the mirrors use the reference's attribute names."""
    from .models import lightstereo as LS, igev_update as UP
    plan = [("stereo.modeling.models.lightstereo.aggregation", LS, ("Aggregation", "MobileV2Residual", "AttentionModule")),
            ("stereo.modeling.models.igev.update", UP, ("ConvGRU", "BasicMotionEncoder", "DispHead", "BasicMultiUpdateBlock")),
            ("stereo.modeling.models.stereobase.gru_blocks", UP, ("ConvGRU", "BasicMotionEncoder", "DispHead", "BasicMultiUpdateBlock"))]
    done = []
    for mod_name, eng_mod, classes in plan:
        try:
            mod = importlib.import_module(mod_name)
        except Exception as ex:
            if verbose:
                print(f"[attach] skip {mod_name}: {type(ex).__name__}: {ex}")
            continue
        for cname in classes:
            if hasattr(mod, cname):
                _graft(getattr(mod, cname), getattr(eng_mod, cname), ("forward", "forward_cl", "_pack", "reset_engine"))
                done.append(f"{mod_name}.{cname}")
    return done


_MISSING = object()


def unpatch_reference():
    while _saved:
        obj, attr, old = _saved.pop()
        if old is _MISSING:
            if attr in getattr(obj, "__dict__", {}):
                delattr(obj, attr)
        else:
            setattr(obj, attr, old)


def attach_gwcnet(model):
    """Swap a reference GwcNet's stages for the engine's, sharing the SAME parameters (state_dict keys
    are identical, so this is a load_state_dict).  Returns an engine GwcNet in eval mode."""
    from .models.gwcnet import GwcNet, _Cfg
    cp, dp = model.CostProcessor, model.DispProcessor
    cfg = _Cfg(MAX_DISP=model.maxdisp, USE_CONCAT_VOLUME=cp.use_concat_volume, CONCAT_CHANNELS=dp.concat_channels,
               DOWNSAMPLE=cp.downsample, NUM_GROUPS=cp.num_groups)
    eng = GwcNet(cfg)
    eng.load_state_dict(model.state_dict(), strict=True)
    dev = next(model.parameters()).device
    return eng.to(dev).eval()
