"""CPU: the patcher rebinds every copy of the hot-path helpers in an unmodified reference checkout
(only runs where /root/reference is mounted) and restores them."""
import os

import pytest

REF = os.environ.get("OPENSTEREO_REF", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not mounted")


def test_patch_and_unpatch_reference():
    import importlib
    from openstereo_amd import attach, ops
    attach.stub_reference_packages(REF)
    cv = importlib.import_module("stereo.modeling.cost_volume.cost_volume")
    sb = importlib.import_module("stereo.modeling.models.gwcnet.gwcnet_disp_processor")
    orig = cv.build_gwc_volume
    done = attach.patch_reference()
    try:
        assert cv.build_gwc_volume is attach.build_gwc_volume and cv.correlation_volume is attach.correlation_volume
        assert "stereo.modeling.cost_volume.cost_volume.build_gwc_volume" in done
        assert "stereo.modeling.models.psmnet.psmnet_cost_processor.cat_fms" in done
        assert "GwcVolumeCostProcessor.build_gwc_volume" in done
        assert sb.disparity_regression is not None and sb.disparity_regression.__name__ == "<lambda>"
        assert len(done) >= 12
    finally:
        attach.unpatch_reference()
    assert cv.build_gwc_volume is orig


def test_patch_reference_modules_grafts_engine_forwards():
    """LightStereo aggregation and the IGEV update block of the REAL reference get the engine forwards
    (class-level graft); on CPU tensors they refuse to run (no CPU path) and unpatch restores the originals."""
    import importlib
    import sys
    import types
    import torch
    from openstereo_amd import attach
    from openstereo_amd.models import lightstereo as LS
    attach.stub_reference_packages(REF)
    agg_mod = importlib.import_module("stereo.modeling.models.lightstereo.aggregation")
    upd_mod = importlib.import_module("stereo.modeling.models.igev.update")
    orig_fwd = agg_mod.Aggregation.forward
    done = attach.patch_reference_modules()
    try:
        assert "stereo.modeling.models.lightstereo.aggregation.Aggregation" in done
        assert "stereo.modeling.models.igev.update.BasicMultiUpdateBlock" in done
        assert agg_mod.Aggregation.forward is not orig_fwd and agg_mod.Aggregation.forward_cl is LS.Aggregation.forward_cl
        assert hasattr(agg_mod.MobileV2Residual, "forward_cl")
        agg = agg_mod.Aggregation(in_channels=48, left_att=True, blocks=[1, 2, 4], expanse_ratio=4,
                                  backbone_channels=[24, 32, 96, 160]).eval()
        with torch.no_grad(), pytest.raises(RuntimeError, match="GPU engine only"):
            agg(torch.zeros(1, 48, 8, 16), [torch.zeros(1, 24, 8, 16), torch.zeros(1, 32, 4, 8), torch.zeros(1, 96, 2, 4)])
        assert upd_mod.ConvGRU._eng is None
    finally:
        attach.unpatch_reference()
    assert agg_mod.Aggregation.forward is orig_fwd and not hasattr(agg_mod.MobileV2Residual, "forward_cl")
