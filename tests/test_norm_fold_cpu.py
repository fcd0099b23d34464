"""CPU: how pack sites classify the normalisation module behind a convolution (engine.norm_kind / foldable_bn).

The reference's trainer converts every BatchNorm to nn.SyncBatchNorm before DDP when SYNC_BN is set (trainer_template.py:83-85; every
BASELINE config sets it).  SyncBatchNorm derives from `_BatchNorm`, not from BatchNorm2d / 3d; three pack sites used to test for the
latter and silently folded nothing for a converted model.  The rule now: any `_BatchNorm` with running statistics
folds, InstanceNorm is its own kernel, nothing / nn.Identity is "no norm", everything else RAISES."""
import pytest
import torch.nn as nn


def test_norm_kind_rules():
    from openstereo_amd import _lib
    from openstereo_amd.engine import foldable_bn, norm_kind
    assert norm_kind(None) is None and norm_kind(nn.Identity()) is None
    for bn in (nn.BatchNorm1d(4), nn.BatchNorm2d(4), nn.BatchNorm3d(4), nn.SyncBatchNorm(4)):
        assert norm_kind(bn) == "bn" and foldable_bn(bn) is bn
    assert norm_kind(nn.InstanceNorm2d(4)) == "in" and norm_kind(nn.InstanceNorm3d(4)) == "in"
    for bad in (nn.GroupNorm(2, 4), nn.LayerNorm(4), nn.ReLU(), nn.BatchNorm2d(4, track_running_stats=False)):
        with pytest.raises(_lib.EngineError):
            norm_kind(bad)
    with pytest.raises(_lib.EngineError):
        foldable_bn(nn.InstanceNorm2d(4))


def test_converted_modules_keep_their_norms_at_the_three_pack_sites():
    """igev_style._pack_sb, context_encoder.ResidualBlock.packs and feature_pyramid._unit with the engine layer replaced by a recorder:
    after convert_sync_batchnorm every site still hands its norm to the conv launch."""
    import torch
    from openstereo_amd.models import context_encoder as CE, feature_pyramid as FP, igev_style as IS
    seen = []

    class Rec:
        def __init__(self, conv, bn=None, act=0, slope=0.01, precision=None):
            seen.append(bn)
            self.Co = conv.out_channels

    sync = nn.SyncBatchNorm.convert_sync_batchnorm
    for mod, name in ((IS, "PackedConv3d"), (CE, "PackedConv3d"), (FP, "PackedConv3d")):
        setattr(mod, "_saved_" + name, getattr(mod, name))
        setattr(mod, name, Rec)
    try:
        blk = sync(IS.BasicConv3d(8, 8, norm_layer=nn.BatchNorm3d, act_layer=nn.LeakyReLU, kernel_size=3, padding=1)).eval()
        IS._pack_sb(blk)
        assert isinstance(seen[-1], nn.SyncBatchNorm)
        rb = sync(CE.ResidualBlock(8, 16, "batch", stride=2)).eval()
        n0 = len(seen)
        rb.packs()
        assert len(seen) == n0 + 3 and all(isinstance(b, nn.SyncBatchNorm) for b in seen[n0:])       # conv1, conv2, downsample
        n0 = len(seen)
        u = FP._unit(sync(IS.BasicConv2d(8, 8, norm_layer=nn.BatchNorm2d, act_layer=nn.LeakyReLU, kernel_size=3, padding=1)).eval())
        assert isinstance(seen[-1], nn.SyncBatchNorm) and not u.inorm and len(seen) == n0 + 1
        u = FP._unit(IS.BasicConv2d(8, 8, norm_layer=nn.InstanceNorm2d, act_layer=nn.LeakyReLU, kernel_size=3, padding=1).eval())
        assert u.inorm and seen[-1] is None
        bad = IS.BasicConv2d(8, 8, norm_layer=nn.BatchNorm2d, act_layer=nn.LeakyReLU, kernel_size=3, padding=1)
        bad.block[1] = nn.GroupNorm(2, 8)
        from openstereo_amd import _lib
        with pytest.raises(_lib.EngineError):
            FP._unit(bad)
    finally:
        for mod, name in ((IS, "PackedConv3d"), (CE, "PackedConv3d"), (FP, "PackedConv3d")):
            setattr(mod, name, getattr(mod, "_saved_" + name))
            delattr(mod, "_saved_" + name)
