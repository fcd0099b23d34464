"""Headline benchmark: stereo-pairs/s, GwcNet-gc forward (inference), 540x960 padded to 544x960, D=192
(BASELINE.json configs[1]) on N MI355X -- one process per GPU, independent pairs, no data-path collective
(weak scaling).

    python bench.py                                   # 1 GPU, default K / W
    python bench.py --gpus 8 --steps 20 --warmup 5    # re-executes itself under torch.distributed.run, 8 ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W           # what the driver does

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline     : the dominant kernel (3x3x3 conv 32->32 @ 48x136x240), timed live with HIP events on the launch
                 stream in an instrumented replay of the same forward; `rooflines` holds the same record for the
                 volume builder (HBM), the 2-D backbone (MFMA), the fused soft-argmin head (HBM) and the classifier
  cpu_baseline : the CPU path on this box's host cores (rank 0, N=1): the REAL reference modules through the import
                 shim when /root/reference is mounted ("reference"), else the oracle restatement ("port");
                 1 warm-up + 3 timed full-size pairs, median, per-stage split
Other workloads (BASELINE configs [2]-[4]; same JSON contract, their own metric string):
  --workload lightstereo_kitti15   correlation volume -> 2-D aggregation -> soft-argmin -> convex upsample, 384x1248
  --workload igev_refine32         geometry-encoding lookup + 3-level ConvGRU update x 32 iterations, 544x960
  --workload stereobase_train      StereoBase cost stage training step (fwd + bwd + SGD), 320x736 crop, DDP over RCCL
  --workload gwcnet_train          GwcNet training step, 256x512 crop, DDP over RCCL
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H_IMG, W_IMG, H_PAD, W_PAD, MAXDISP = 540, 960, 544, 960, 192
# MI355X_MICROARCH.md dense peaks.  f16x3 executes 3 fp16 MFMAs per fp32-equivalent product, so the roofline for
# ALGORITHMIC flops in that mode is 2500 / 3.
PEAKS = {"f32": (157.3, "v_mfma_f32_32x32x2_f32 dense peak"),
         "f16x3": (2500.0 / 3.0, "fp16 MFMA dense peak 2500 TF / 3 MFMAs per fp32-equivalent product"),
         "f16": (2500.0, "fp16 MFMA dense peak (one MFMA per product, fp32 accumulate)")}
HBM_PEAK = 8000.0          # GB/s (spec; ~6.3 TB/s achievable)
DTYPES = {"f32": "f32", "f16x3": "f32 via f16x3 split-MFMA (hi/lo fp16 operands, f32 accumulate; HBM tensors f32)",
          "f16": "f16 (the reference's autocast arithmetic: fp16 operands, one MFMA per product, f32 accumulate and epilogue)"}
# algorithmic work per pair
DOM_GFLOP = 2 * 27 * 32 * 32 * 48 * 136 * 240 / 1e9       # one 3x3x3 32->32 layer at 48x136x240 (43.32 GMAC)
BACKBONE_GFLOP = 461.6
VOLUME_MB, HEAD_MB, CLASSIF_MB = 487.8, 8.36, 200.5 + 6.27
# rocprofv3 names of the timed kernels (verbatim, as in profiles/round2/*_kernel_stats.csv)
KERNEL_NAMES = {
    ("conv", "f16x3"): "void osa::conv_march_kernel<4, 16, 1, 1>(osa::ConvArgs, int, int)   [d-marching form, csrc/conv_march.h]",
    ("conv", "f32"): "void osa::conv_mfma_kernel<0, 1, 1, 2, 1, 4, 1, 8, 8, 0, 0, 0, 1, 0>(osa::ConvArgs)",
    "volume": "void osa::build_volume_walk_kernel<2, 8, 4, true>(osa::VolQArgs)   [d-walking form, split output for the f16x3 chain; fp32 output: <2, 8, 8, false>; csrc/volume.hip]",
    "head": "osa::upsample4_softargmin_kernel(osa::UpArgs)",
    "classifier": "osa::classifier_march_kernel(osa::ConvArgs, float const*, float const*, int, int)",
}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pin_host_threads(local_rank, world):
    """N ranks on one host: each rank keeps to its own contiguous share of the CPUs this job may use (cores are
    enumerated socket by socket, so a contiguous block stays on one NUMA node for the usual 2-socket x 4-GPU layout) and sizes torch's
    intra-op pool to it -- eight ranks would otherwise each start a 128-thread pool and migrate across sockets while they launch kernels.
    World size 1 (the driver's N = 1 run, whose rank 0 also times the CPU baseline on ALL host cores) is left alone.  Returns the CPU set."""
    if world <= 1 or not hasattr(os, "sched_getaffinity"):
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = max(1, len(cpus) // world)
        mine = cpus[local_rank * per:(local_rank + 1) * per] or cpus[-per:]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), 16)))
        return mine
    except OSError:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU per step (default: the best-throughput batch of the sweeps in DESIGN.md 6 -- 9 GwcNet, 8 LightStereo, 4 IGEV, 1 training; single-pair latency is reported next to it)")
    ap.add_argument("--streams", type=int, default=None, help="GwcNet inference: independent sub-batches on this many concurrent HIP streams (1 = one stream; default: 3 when the batch divides by 3 -- sub-batches of 3 pairs measured best, profiles/round4/substreams_x_batch_sweep.txt --, else 2, else 1)")
    ap.add_argument("--workload", default="gwcnet",
                    choices=("gwcnet", "lightstereo_kitti15", "igev_refine32", "stereobase_train", "stereobase_e2e_train", "gwcnet_train",
                             "stereobase_e2e", "igev_e2e", "lightstereo_e2e"))
    ap.add_argument("--precision", choices=("f16x3", "f32", "f16"), default=os.environ.get("OSA_PRECISION", "f16x3"),
                    help="MFMA arithmetic mode of the conv kernels (both pass the same parity tests)")
    ap.add_argument("--stages", default="", help="write the full per-stage timing table to this file")
    ap.add_argument("--amp", action="store_true", help="run the step the way the reference runs its AMP configs: inference inside torch.autocast(fp16); training as autocast + GradScaler (trainer_template.py:211-226) -- engine layers in the native f16 mode")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (default workload) / the same-GPU PyTorch-ROCm eager leg (training workloads)")
    ap.add_argument("--no-workloads", action="store_true", help="default workload: skip the compact measurements of the other BASELINE configs (`workloads`)")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling runs: nothing but warm-up + the timed configuration (no other-precision leg, no B=1 loop, no CPU leg)")
    ap.add_argument("--force-ddp", action="store_true", help="training workloads on ONE GPU: one-rank RCCL process group + DistributedDataParallel (checks the DDP + hipGraph path without a second GPU)")
    ap.add_argument("--stub", action="store_true",
                    help="CPU plumbing self-test (tests/test_sharding_gloo.py): gloo backend, the forward replaced by a sleep")
    return ap.parse_args()


# ============================================================================================ workloads
class GwcNetInference:
    metric = "stereo-pairs/s at 540x960 D=192 (GwcNet fwd)"
    scaling, graphable, training = "weak", True, False

    def __init__(self, args, dev, rank):
        from openstereo_amd.models.gwcnet import GwcNet
        from openstereo_amd.utils.weights import synth_state_dict, synth_images
        self.B = args.batch or 9
        net = GwcNet()
        self.sd = synth_state_dict(net, seed=0)
        net.load_state_dict(self.sd)
        self.net = net.to(dev).eval()
        # 540x960 SceneFlow-shaped pair, edge-padded top/right to 544x960 (RightTopPad, stereo_trans.py:243-267)
        L0, R0 = synth_images(self.B, H_IMG, W_IMG, seed=1 + rank)
        pad = lambda t: torch.nn.functional.pad(t, (0, W_PAD - W_IMG, H_PAD - H_IMG, 0), mode="replicate")
        self.L, self.R = pad(L0).to(dev), pad(R0).to(dev)        # inputs resident in HBM before timing starts
        from openstereo_amd.parallel import SubBatchStreams
        want = args.streams if args.streams else (3 if self.B % 3 == 0 else (2 if self.B % 2 == 0 else 1))
        self.nstreams = want if self.B % want == 0 else 1
        self.sub = SubBatchStreams(self.nstreams)                # independent sub-batches on concurrent HIP streams (fork / join inside the hipGraph)

    def step(self):
        with torch.no_grad():
            return self.sub(lambda L, R: self.net({"left": L, "right": R})["disp_pred"], self.L, self.R)

    def step_single(self):
        """the same forward as ONE launch sequence over all B pairs (the roofline leg times kernels one at a time)"""
        with torch.no_grad():
            return self.net({"left": self.L, "right": self.R})["disp_pred"]

    def config(self, args):
        return {"workload": "GwcNet-gc inference, SceneFlow-shaped 540x960 padded to 544x960, D=192, G=40 + 12ch concat "
                            "(BASELINE configs[1])",
                "weights": "deterministic synthetic (sharpened), random-init architecture",
                "sub_batch_streams": self.nstreams}


class LightStereoKitti15:
    """BASELINE configs[3]: LightStereo-S hot path at KITTI15 size (375x1242 padded to 384x1248, cfgs/lightstereo/kitti15_eval.yaml:12):
    correlation_volume -> Aggregation -> softmax + disparity_regression -> context_upsample (lightstereo.py:51-62).  The timm
    feature extractor is not available offline: its outputs are synthetic NCHW feature maps of the documented shapes."""
    metric = "stereo-pairs/s, LightStereo-S cost stage at 384x1248 D=192"
    scaling, graphable, training = "weak", True, False

    def __init__(self, args, dev, rank):
        from openstereo_amd.models.lightstereo import LightStereoCostStage
        from openstereo_amd.utils.weights import synth_state_dict
        self.B = B = args.batch or 8
        st = LightStereoCostStage(max_disp=192)
        st.load_state_dict(synth_state_dict(st, seed=9))
        self.st = st.to(dev).eval()
        g = torch.Generator().manual_seed(60 + rank)
        r = lambda *s: torch.randn(*s, generator=g).to(dev)
        self.fl = [r(B, 24, 96, 312), r(B, 32, 48, 156), r(B, 96, 24, 78)]
        self.fr = torch.roll(self.fl[0], -3, 3) + 0.1 * r(B, 24, 96, 312)
        self.spx = r(B, 9, 384, 1248)

    def step(self):
        from openstereo_amd import ops
        with torch.no_grad():
            out = self.st(self.fl, self.fr)
            return ops.context_upsample(out["init_disp"], self.spx, softmax_weights=True, gain=4.0)

    def config(self, args):
        return {"workload": "LightStereo-S: correlation volume (24ch, D/4=48) -> 2-D aggregation -> soft-argmin -> convex x4 upsample, "
                            "KITTI15 375x1242 padded to 384x1248 (BASELINE configs[3]); synthetic feature maps stand in for the timm backbone"}


class IGEVRefine32:
    """BASELINE configs[4]: the IGEV refinement loop at SceneFlow size (quarter resolution 136x240), VALID_ITERS = 32
    (cfgs/igev/igev_sceneflow_amp.yaml:31; igev_stereo.py:181-203) + the final convex upsample."""
    metric = "stereo-pairs/s, IGEV GRU refinement x32 at 544x960"
    scaling, graphable, training = "weak", True, False      # ~1500 launches per pair: launch-bound unless replayed as a hipGraph

    def __init__(self, args, dev, rank):
        from openstereo_amd.models.igev_update import IGEVRefiner
        from openstereo_amd.utils.weights import synth_state_dict
        self.B = B = args.batch or 4
        a = _Cfg(CORR_LEVELS=2, CORR_RADIUS=4, N_GRU_LAYERS=3, N_DOWNSAMPLE=2, SLOW_FAST_GRU=True)
        ref = IGEVRefiner(a, hidden_dims=[128, 128, 128])
        ref.load_state_dict(synth_state_dict(ref, seed=11))
        self.ref = ref.to(dev).eval()
        g = torch.Generator().manual_seed(70 + rank)
        r = lambda *s: torch.randn(*s, generator=g).to(dev)
        H, W = 136, 240
        self.ml, self.mr, self.gvol = r(B, 96, H, W), r(B, 96, H, W), r(B, 8, 48, H, W)
        self.net = [torch.tanh(r(B, 128, H >> i, W >> i)) for i in range(3)]
        self.inp = [[0.5 * r(B, 128, H >> i, W >> i) for _ in range(3)] for i in range(3)]
        self.d0 = r(B, 1, H, W).abs() * 3
        self.spx = r(B, 9, 4 * H, 4 * W)

    def step(self):
        from openstereo_amd import ops
        with torch.no_grad():
            out = self.ref(self.ml, self.mr, self.gvol, list(self.net), self.inp, self.d0, 32)
            return ops.context_upsample(out["disp"], self.spx, softmax_weights=True, gain=4.0)

    def config(self, args):
        return {"workload": "IGEV refinement: geometry-encoding volume (all-pairs corr + pyramid) once, then 32 x (fused 2-level 9-tap lookup "
                            "+ 3-level ConvGRU update block + disparity update) at 136x240, + convex x4 upsample to 544x960 "
                            "(BASELINE configs[4]); synthetic features / hidden states stand in for the timm extractor"}


def _amp_training_step(wl, forward_loss, clip=None):
    """One optimisation step the way the reference's trainer runs it (trainer_template.py:202-226): zero_grad; forward + loss under
    `torch.autocast(enabled=AMP)`; `scaler.scale(loss).backward()`; `scaler.unscale_`; gradient clipping; `scaler.step`; `scaler.update`.
    wl.amp False: the plain fp32-class step (autocast and the scaler disabled: identical to the previous rounds' step).  With AMP the
    optimizers are the fused variants, which take the scaler's found-inf flag on the device (no host synchronisation: the whole step
    stays capturable in a hipGraph)."""
    wl.opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=wl.amp):
        loss = forward_loss()
    if not wl.amp:
        loss.backward()
        if clip is not None:
            clip()
        wl.opt.step()
        return loss.detach()
    wl.scaler.scale(loss).backward()
    wl.scaler.unscale_(wl.opt)
    if clip is not None:
        clip()
    wl.scaler.step(wl.opt)
    wl.scaler.update()
    return loss.detach().float()


def _set_amp(wl, args):
    wl.amp = bool(getattr(args, "amp", False))
    # (init_scale 2^12: the synthetic losses here have gradients around 1e-3 .. 1e-1; the default 2^16 is also fine, it just spends the
    # first steps of a run backing off)
    wl.scaler = torch.amp.GradScaler("cuda", enabled=wl.amp, init_scale=4096.0)
    return dict(fused=True) if wl.amp else {}


class StereoBaseTrain:
    """BASELINE configs[2]: StereoBase training step on the hot path -- gwc(8) + concat volume -> Hourglass(24) with FeatureAtt ->
    classifier -> softmax regression (stereobase_gru.py:139-164), forward + backward + SGD, SceneFlow crop 320x736
    (cfgs/stereobase/stereobase_sceneflow.yaml:15-16), data parallel with DistributedDataParallel (RCCL all-reduce)."""
    metric = "training stereo-pairs/s, StereoBase cost stage at 320x736 crop"
    scaling, graphable, training = "weak", False, True

    def __init__(self, args, dev, rank):
        from openstereo_amd.models.igev_style import StereoBaseCostStage
        from openstereo_amd.utils.weights import synth_state_dict
        self.B = B = args.batch or 1
        st = StereoBaseCostStage(max_disp=192, num_groups=8, concat_channels=8, backbone_channels=[48, 64, 192, 120])
        st.load_state_dict(synth_state_dict(st, seed=8, head_gain=20.0))
        st = st.to(dev).train()
        for m in st.modules():                                    # FREEZE_BN: true (stereobase_sceneflow.yaml:48)
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                m.eval()
        self.raw = st
        self.model = st
        if _use_ddp():
            self.model = torch.nn.parallel.DistributedDataParallel(st, device_ids=[dev.index])
        self.opt = torch.optim.SGD([p for p in st.parameters() if p.requires_grad], lr=1e-4, **_set_amp(self, args))
        g = torch.Generator().manual_seed(80 + rank)
        r = lambda *s: torch.randn(*s, generator=g).to(dev)
        H, W = 80, 184
        self.x = [r(B, 96, H, W).requires_grad_(), r(B, 96, H, W).requires_grad_(), r(B, 8, H, W), r(B, 8, H, W)]
        self.feats = [None, r(B, 64, H // 2, W // 2), r(B, 192, H // 4, W // 4), r(B, 120, H // 8, W // 8)]
        self.gt = torch.rand(B, 1, H, W, generator=g).to(dev) * 40

    static = False            # (the loss of this workload has static shapes already)

    def step(self):
        return _amp_training_step(self, lambda: torch.nn.functional.smooth_l1_loss(self.model(*self.x, self.feats)["init_disp"].float(), self.gt))

    def config(self, args):
        return {"workload": "StereoBase cost stage training step (volume -> Hourglass(24)+FeatureAtt -> classifier -> softmax regression; "
                            "fwd + bwd + SGD, frozen BN), SceneFlow crop 320x736, 1 pair per GPU (BASELINE configs[2]); synthetic feature "
                            "maps stand in for the timm backbone"}


class _E2EInference:
    """Whole-model inference of the end-to-end classes of openstereo_amd/models/stereo_models.py with the reference's feature pyramid
    (feature_pyramid.py: MobileNetV2-100 trunk mirror -- unpinned, timm absent -- + the reference-written FPN decoder, pinned), all of it
    on the engine in eval mode."""
    scaling, graphable, training = "weak", True, False
    H, W = 544, 960

    def _finish(self, net, args, dev, rank, seed, scale255=False):
        from openstereo_amd.utils.weights import synth_state_dict, synth_images
        net.load_state_dict(synth_state_dict(net, seed=seed, head_gain=20.0, gain=0.9))
        self.net = net.to(dev).eval()
        L, R = synth_images(self.B, self.H, self.W, seed=30 + rank)
        if scale255:
            L, R = (L * 40 + 128).clamp(0, 255), (R * 40 + 128).clamp(0, 255)
        self.L, self.R = L.to(dev), R.to(dev)

    def step(self):
        with torch.no_grad():
            return self.net({"left": self.L, "right": self.R})["disp_pred"]


class StereoBaseE2E(_E2EInference):
    metric = "stereo-pairs/s, StereoBase (MobileNetV2-100 trunk mirror, unpinned + reference FPN, pinned) at 544x960 D=192, 32 GRU iterations"

    def __init__(self, args, dev, rank):
        from types import SimpleNamespace
        from openstereo_amd.models.stereo_models import StereoBase
        self.B = args.batch or 2
        cfg = SimpleNamespace(MAX_DISP=192, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128],
                              N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=False, EVAL_ITERS=32, TRAIN_ITERS=22)
        self._finish(StereoBase(cfg, feature="mobilenetv2"), args, dev, rank, 41)

    def config(self, args):
        return {"workload": "StereoBase inference, whole model incl. the MobileNetV2-100 feature pyramid on the engine (gwc + concat volume -> hourglass -> classifier -> regression -> "
                            "geometry lookup + 32 GRU iterations -> convex upsampling), 544x960 D=192 (cfgs/stereobase/stereobase_sceneflow.yaml, EVAL_ITERS 32)"}


class IGEVE2E(_E2EInference):
    metric = "stereo-pairs/s, IGEV-Stereo (MobileNetV2-100 trunk mirror, unpinned + reference FPN, pinned) at 544x960 D=192, 32 GRU iterations"

    def __init__(self, args, dev, rank):
        from types import SimpleNamespace
        from openstereo_amd.models.stereo_models import IGEVStereo
        self.B = args.batch or 2
        a = SimpleNamespace(MAX_DISP=192, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=True,
                            VALID_ITERS=32, N_DOWNSAMPLE=2)
        self._finish(IGEVStereo(a, feature="mobilenetv2"), args, dev, rank, 43, scale255=True)

    def config(self, args):
        return {"workload": "IGEV-Stereo inference, whole model incl. the MobileNetV2-100 feature pyramid on the engine (gwc volume -> corr_stem + FeatureAtt -> hourglass -> classifier -> "
                            "regression -> geometry lookup + 32 slow-fast GRU iterations -> convex upsampling), 544x960 D=192 (BASELINE configs[4])"}


class LightStereoE2E(_E2EInference):
    metric = "stereo-pairs/s, LightStereo-S (MobileNetV2-100 trunk mirror, unpinned + reference FPN, pinned) at 384x1248 D=192"
    H, W = 384, 1248

    def __init__(self, args, dev, rank):
        from types import SimpleNamespace
        from openstereo_amd.models.stereo_models import LightStereo
        self.B = args.batch or 8
        cfg = SimpleNamespace(MAX_DISP=192, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)
        self._finish(LightStereo(cfg, backbone="mobilenetv2"), args, dev, rank, 47)

    def config(self, args):
        return {"workload": "LightStereo-S inference, whole model incl. the MobileNetV2-100 feature pyramid on the engine (correlation volume -> 2-D aggregation -> regression -> convex "
                            "upsampling), KITTI15 375x1242 padded to 384x1248 (BASELINE configs[3])"}


class StereoBaseE2ETrain:
    """BASELINE configs[2], whole model: openstereo_amd.models.stereo_models.StereoBase in training mode at the SceneFlow crop 320x736
    (cfgs/stereobase/stereobase_sceneflow.yaml) -- volumes, hourglass, classifier, regression, geometry-encoding lookup and 22 GRU iterations
    (TRAIN_ITERS) with convex upsampling after each, the loss of stereobase_gru.py:215-243, backward, AdamW step; frozen BN.  The timm
    pyramid / context encoder are the shape-compatible stand-ins (torch modules, trained along)."""
    metric = "training stereo-pairs/s, StereoBase (MobileNetV2-100 trunk mirror + reference FPN) at 320x736 crop, 22 GRU iterations"
    scaling, graphable, training = "weak", False, True

    def __init__(self, args, dev, rank):
        import numpy as np
        from types import SimpleNamespace
        from openstereo_amd.models.stereo_models import StereoBase
        from openstereo_amd.utils.weights import synth_state_dict, synth_images
        self.B = B = args.batch or 1
        cfg = SimpleNamespace(MAX_DISP=192, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128],
                              N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=False, EVAL_ITERS=32, TRAIN_ITERS=22)
        net = StereoBase(cfg, feature="mobilenetv2")
        net.load_state_dict(synth_state_dict(net, seed=41, head_gain=20.0, gain=0.9))
        net = net.to(dev).train()
        for m in net.modules():                                   # FREEZE_BN: true
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                m.eval()
        self.raw = net
        self.model = net
        if _use_ddp():
            self.model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[dev.index])
        # capturable: the step counter lives on the device, so the optimizer step can be part of a hipGraph (no effect on the arithmetic)
        self.opt = torch.optim.AdamW([p for p in net.parameters() if p.requires_grad], lr=2e-4, weight_decay=1e-5, eps=1e-8, capturable=True, **_set_amp(self, args))
        L, R = synth_images(B, 320, 736, seed=20 + rank)
        self.L, self.R = L.to(dev), R.to(dev)
        self.gt = torch.from_numpy(np.random.default_rng(rank).uniform(1, 100, (B, 320, 736)).astype("float32")).to(dev)

    static = False            # True: static-shape loss without .item() (hipGraph replay of the whole step)

    def step(self):
        return _amp_training_step(self, lambda: self.raw.get_loss(self.model({"left": self.L, "right": self.R}), {"disp": self.gt}, static=self.static)[0],
                                  clip=lambda: torch.nn.utils.clip_grad_value_(self.raw.parameters(), 1.0))          # CLIP_GRAD: value 1.0

    def config(self, args):
        return {"workload": "StereoBase training step, whole model with stand-in 2-D backbone (cost stage + geometry lookup + 22 GRU iterations + convex "
                            "upsampling; fwd + bwd + AdamW, frozen BN, grad clip), SceneFlow crop 320x736, 1 pair per GPU (BASELINE configs[2])"}


class GwcNetTrain:
    metric = "training stereo-pairs/s, GwcNet at 256x512 crop"
    scaling, graphable, training = "weak", False, True

    def __init__(self, args, dev, rank):
        import numpy as np
        from openstereo_amd.models.gwcnet import GwcNet
        from openstereo_amd.utils.weights import synth_state_dict, synth_images
        self.B = B = args.batch or 1
        net = GwcNet()
        net.load_state_dict(synth_state_dict(net, seed=0))
        self.raw = net.to(dev).train()
        self.model = self.raw
        if _use_ddp():
            self.model = torch.nn.parallel.DistributedDataParallel(self.raw, device_ids=[dev.index])
        self.amp, self.scaler = False, None                   # cfgs/gwcnet/gwcnet_sceneflow.yaml: AMP false
        self.opt = torch.optim.RMSprop(self.model.parameters(), lr=1e-3, capturable=True)      # cfgs/gwcnet/gwcnet_sceneflow.yaml (capturable: hipGraph-friendly, same arithmetic)
        L, R = synth_images(B, 256, 512, seed=10 + rank)
        self.L, self.R = L.to(dev), R.to(dev)
        self.gt = torch.from_numpy(np.random.default_rng(rank).uniform(1, 100, (B, 256, 512)).astype("float32")).to(dev)

    static = False

    def step(self):
        self.opt.zero_grad(set_to_none=True)
        out = self.model({"left": self.L, "right": self.R})
        loss, _ = self.raw.get_loss(out, {"disp": self.gt}, static=self.static)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def config(self, args):
        return {"workload": "GwcNet-gc training step (4 supervised heads, fwd + bwd + RMSprop, batch statistics BN), SceneFlow crop 256x512 "
                            "(cfgs/gwcnet/gwcnet_sceneflow.yaml), 1 pair per GPU"}


class _Stub:
    """--stub: no GPU, no engine; exercises launcher, process group, barrier, MAX-over-ranks timing and the JSON line."""
    metric = "stub-pairs/s"
    scaling, graphable, training = "weak", False, False

    def __init__(self, args, dev, rank):
        self.B, self.rank = args.batch or 2, rank

    def step(self):
        time.sleep(0.002 * (1 + self.rank))
        return torch.zeros(1)

    def config(self, args):
        return {"workload": "stub"}


class _Cfg(dict):
    __getattr__ = dict.__getitem__


WORKLOADS = {"gwcnet": GwcNetInference, "lightstereo_kitti15": LightStereoKitti15, "igev_refine32": IGEVRefine32,
             "stereobase_train": StereoBaseTrain, "stereobase_e2e_train": StereoBaseE2ETrain, "gwcnet_train": GwcNetTrain,
             "stereobase_e2e": StereoBaseE2E, "igev_e2e": IGEVE2E, "lightstereo_e2e": LightStereoE2E}


# ============================================================================================ rooflines (GwcNet)
def gwcnet_rooflines(wl, args, eager_step, nrep):
    """Instrumented replay: HIP events around every engine launch on the launch stream."""
    from openstereo_amd import engine
    B, prec = wl.B, args.precision
    rec = engine.enable_timing()
    for _ in range(nrep):
        eager_step()
    torch.cuda.synchronize()
    stats = engine.collect_timing(rec)
    per_step = {"/".join(map(str, k)): round(sum(v) / nrep, 4) for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1]))}
    traffic = {}
    tj = os.path.join(ROOT, "profiles", "traffic.json")                    # PMC bytes per launch, measured offline (profiles/round2)
    if os.path.exists(tj):
        traffic = json.load(open(tj))

    def avg_ms(pred):
        sel = [v for k, v in stats.items() if pred(k)]
        n = sum(len(v) for v in sel)
        return (sum(sum(v) for v in sel) / n) if n else None

    peak, why = PEAKS[prec]
    out = []
    ms = avg_ms(lambda k: k[0] == "conv3d" and k[1:] == (32, 32, 3, 1, 48, 136, 240))
    if ms:
        ach = DOM_GFLOP * B / ms
        out.append({"kernel": KERNEL_NAMES[("conv", prec)], "what": f"3x3x3 conv 32->32 @48x136x240, {B} pairs per launch (4 launches per step)" + (", d-marching form" if prec == "f16x3" else ""),
                    "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "peak_note": why, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "algorithmic_gflop_per_launch": round(DOM_GFLOP * B, 2),
                    "traffic": traffic.get(f"conv3d_32_32_V0_{prec}_B{B}"), "avg_launch_ms": round(ms, 4)})
    ms = avg_ms(lambda k: k[0] == "build_volume")
    if ms:
        ach = VOLUME_MB * B / ms                                             # MB / ms = GB/s
        out.append({"kernel": KERNEL_NAMES["volume"], "what": f"fused gwc(40)+concat(24) volume, NDHWC, {B} pairs per launch", "bound": "hbm",
                    "achieved": round(ach, 1), "peak": HBM_PEAK, "unit": "GB/s", "frac": round(ach / HBM_PEAK, 4),
                    "algorithmic_mb_per_launch": round(VOLUME_MB * B, 1), "traffic": traffic.get(f"volume_B{B}"), "avg_launch_ms": round(ms, 4)})
    # the backbone's GPU time = the SUM of its launches' own event pairs (every D == 1 convolution of this forward belongs to it).  The span
    # around the whole backbone (stage `backbone2d_engine`) also contains the host's launch gaps of this instrumented eager replay -- 170 event
    # records for 85 launches -- and read 28 ms instead of 17 on a box with a slow host (profiles/round4/bench_driver_style_stdout.txt).
    span_ms = avg_ms(lambda k: k[0] == "backbone2d_engine")
    ms = sum(sum(v) for k, v in stats.items() if k[0] in ("conv3d", "deconv3d") and len(k) >= 8 and k[5] == 1) / nrep
    if ms:
        ach = BACKBONE_GFLOP * B / ms
        out.append({"kernel": "2-D feature extractor: ~85 launches of osa::conv_mfma_kernel<...> with D = 1 (both images of every pair)",
                    "what": f"GwcNet backbone, {2 * B} images per step: sum of its launches' event-pair durations", "bound": "mfma", "achieved": round(ach, 2),
                    "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None, "avg_launch_ms": round(ms, 4),
                    "enclosing_span_ms": None if span_ms is None else round(span_ms, 4)})
    ms = avg_ms(lambda k: k[0] == "upsample_softargmin")
    if ms:
        # the fused head reads 8 MB per pair and evaluates one exp per (disparity, pixel) sample: it sits under the transcendental-issue
        # roof (v_exp_f32 is quarter rate: 256 CUs x 4 SIMDs x 16 lanes / 4 per clock at 2.4 GHz), not under HBM
        samples = MAXDISP * H_PAD * W_PAD * B
        exp_peak = 256 * 4 * 16 / 4 * 2.4e9 / 1e12                           # T exp/s
        ach = samples / ms / 1e9                                              # T samples/s
        out.append({"kernel": KERNEL_NAMES["head"], "what": f"fused trilinear x4 + softmax + expectation, {B} pairs per launch", "bound": "valu-exp",
                    "achieved": round(ach, 3), "peak": round(exp_peak, 2), "unit": "T samples/s (one v_exp_f32 each; quarter-rate issue)", "frac": round(ach / exp_peak, 4),
                    "hbm_gb_s": round(HEAD_MB * B / ms, 1), "hbm_frac": round(HEAD_MB * B / ms / HBM_PEAK, 4),
                    "traffic": traffic.get(f"head_B{B}"), "avg_launch_ms": round(ms, 4)})
    ms = avg_ms(lambda k: k[0] == "conv3d_small_co")
    if ms:
        ach = CLASSIF_MB * B / ms
        out.append({"kernel": KERNEL_NAMES["classifier"], "what": f"classif3.2: 3x3x3 conv 32->1 @48x136x240, {B} pairs per launch", "bound": "hbm",
                    "achieved": round(ach, 1), "peak": HBM_PEAK, "unit": "GB/s", "frac": round(ach / HBM_PEAK, 4),
                    "traffic": traffic.get(f"classifier_B{B}"), "avg_launch_ms": round(ms, 4)})
    return out, per_step


def generic_rooflines(wl, args, eager_step, nrep):
    """Workloads other than GwcNet inference: instrumented replay, per-stage table, and a roofline record for the convolution launch
    class that takes the most time (algorithmic flops / bytes of one launch are recorded by the engine next to each timed span)."""
    from openstereo_amd import engine, timing
    rec = engine.enable_timing()
    for _ in range(nrep):
        eager_step()
    torch.cuda.synchronize()
    stats = engine.collect_timing(rec)
    per_step = {"/".join(map(str, k)): round(sum(v) / nrep, 4) for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1]))}
    peak, why = PEAKS[args.precision]
    convs = {k: v for k, v in stats.items() if k in timing.work and k[0] in ("conv3d", "deconv3d", "wgrad", "wgrad_f16x3")}
    out = []
    for k, v in sorted(convs.items(), key=lambda kv: -sum(kv[1]))[:3]:
        ms = sum(v) / len(v)
        flops, nbytes = timing.work[k]
        tf, gbs = flops / ms / 1e9, nbytes / ms / 1e6
        kind, ci, co, kk, st, d, h, w = k[:8]
        if kind == "wgrad":                                  # exact-fp32 MFMA weight gradients (strided / transposed layers; every layer in f32 mode)
            peak, why = PEAKS["f32"]
        elif kind == "wgrad_f16x3":                          # split-precision weight gradients (csrc/wgrad.hip, wgrad_f16x3_kernel)
            peak, why = PEAKS["f16x3"]
        else:
            peak, why = PEAKS[args.precision]
        mfma_bound = flops / (peak * 1e12) >= nbytes / (HBM_PEAK * 1e9)              # which roofline the launch sits under
        rec_ = {"kernel": "osa::wgrad_kernel<...> + osa::wgrad_reduce_kernel (csrc/wgrad.hip)" if kind == "wgrad" else
                          ("osa::wgrad_f16x3_kernel<...> + osa::wgrad_reduce_kernel (csrc/wgrad.hip)" if kind == "wgrad_f16x3" else
                           "osa::conv_mfma_kernel<...> (tile picked per layer, csrc/conv3d.hip pick_cfg)"),
                "what": f"{kind} {ci}->{co} k{kk} stride {st} @ {d}x{h}x{w}, {len(v) // nrep} launches per step",
                "bound": "mfma" if mfma_bound else "hbm", "avg_launch_ms": round(ms, 4), "traffic": None,
                "algorithmic_gflop_per_launch": round(flops / 1e9, 3), "algorithmic_mb_per_launch": round(nbytes / 1e6, 2)}
        if mfma_bound:
            rec_.update({"achieved": round(tf, 2), "peak": round(peak, 1), "peak_note": why, "unit": "TFLOP/s", "frac": round(tf / peak, 4)})
        else:
            rec_.update({"achieved": round(gbs, 1), "peak": HBM_PEAK, "unit": "GB/s", "frac": round(gbs / HBM_PEAK, 4)})
        out.append(rec_)
    return out, per_step


# ============================================================================================ same-GPU eager baseline + secondary workloads
class _eager_torch_mode:
    """Baseline leg only: inside this context the engine's autograd entry points are rebound to the plain torch ops the reference's own
    code executes (the oracle restatements / torch.nn.functional), so the SAME workload classes measure what stock PyTorch-ROCm eager
    (MIOpen / rocBLAS) does on this GPU -- training included.  Never active in a timed engine region, never shipped behaviour."""

    def __enter__(self):
        import contextlib
        import torch.nn.functional as F
        from oracle import torch_ref as O                 # baseline leg only
        from openstereo_amd import autograd as AG, geometry as GEO
        self.saved = [(AG, n, getattr(AG, n)) for n in ("engine_convs", "conv_module", "conv3d", "conv2d", "conv_transpose3d", "conv_transpose2d",
                                                         "build_gwc_volume", "build_concat_volume", "correlation_volume", "disparity_regression",
                                                         "softmax_disparity_regression", "upsample_softargmin")]
        self.saved.append((GEO, "CombinedGeoEncodingVolume", GEO.CombinedGeoEncodingVolume))
        AG.engine_convs = contextlib.nullcontext
        AG.conv_module = lambda m, x: m(x)
        AG.conv3d = lambda x, w, b=None, stride=1, padding=0, dilation=1, precision=None: F.conv3d(x, w, b, stride, padding, dilation)
        AG.conv2d = lambda x, w, b=None, stride=1, padding=0, dilation=1, precision=None: F.conv2d(x, w, b, stride, padding, dilation)
        AG.conv_transpose3d = lambda x, w, b=None, stride=2, padding=1, output_padding=0, precision=None: F.conv_transpose3d(x, w, b, stride, padding, output_padding)
        AG.conv_transpose2d = lambda x, w, b=None, stride=2, padding=1, output_padding=0, precision=None: F.conv_transpose2d(x, w, b, stride, padding, output_padding)
        AG.build_gwc_volume = O.gwc_volume
        AG.build_concat_volume = O.concat_volume
        AG.correlation_volume = O.corr_volume
        AG.disparity_regression = lambda p, maxdisp, keepdim=True: O.disparity_regression(p, maxdisp, keepdim)
        AG.softmax_disparity_regression = lambda c, keepdim=True: O.disparity_regression(F.softmax(c, 1), c.shape[1], keepdim)
        AG.upsample_softargmin = lambda c, maxdisp, h, w, align_corners=False: O.upsample_regression(c if c.dim() == 5 else c[:, None], maxdisp, h, w, align_corners)

        class TorchGeo:
            def __init__(self, f1, f2, gv, num_levels=2, radius=4):
                self.o, self.meta = O.GeoEncodingVolume(f1.float(), f2.float(), gv.float(), num_levels=num_levels, radius=radius), None

            def __call__(self, disp, coords):
                with torch.device(disp.device):
                    return self.o(disp, coords)
        GEO.CombinedGeoEncodingVolume = TorchGeo
        # r6: the fused training ops that are not reached through the entry points above take their torch compositions too -- the ConvGRU gate
        # kernels (update.py:36-45 as torch ops) and the fused softmax + convex up-sampling (F.softmax + unfold / nearest / weighted sum)
        from openstereo_amd.models import igev_update as IU, stereo_models as SM
        for mod, name in ((IU, "FUSED_GRU_TRAIN"), (SM, "FUSED_UPSAMPLE_TRAIN")):
            self.saved.append((mod, name, getattr(mod, name)))
            setattr(mod, name, False)
        return self

    def __exit__(self, *exc):
        for obj, name, val in self.saved:
            setattr(obj, name, val)
        return False


def _use_ddp():
    """Training workloads wrap their model in DistributedDataParallel when the job has more than one rank -- or when --force-ddp asks
    for the one-rank RCCL group that lets a single-GPU box exercise the DDP + hipGraph path."""
    return int(os.environ.get("WORLD_SIZE", 1)) > 1 or os.environ.get("OSA_BENCH_FORCE_DDP") == "1"


def capture_training_step(wl, ddp=False):
    """Whole training step as ONE hipGraph (forward, loss, backward, optimizer step, the per-step weight re-packs with their device-side
    scales): the eager step is launch-bound (1.7 K - 11 K launches of a few microseconds each).  Nothing is skipped -- every replay runs
    the same kernels on the updated weights; the loss is the static-shape form (`get_loss(..., static=True)`, same value).  PyTorch's
    whole-network capture recipe: warm-up on a side stream, grads released before capture so that backward allocates them from the
    graph's pool.  Under DDP (r4) the recipe's extra conditions hold too: the wrapper was constructed on a side stream (main), the RCCL
    watchdog's asynchronous error handling is off (set before init_process_group), and 11 eager DDP steps precede the capture (the reducer
    rebuilds its buckets during the first iterations); the gradient all-reduces are then captured as graph nodes on RCCL's stream.
    Returns (graph, step) or None when capture is not possible (the caller then times eager steps)."""
    try:
        wl.static = True
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(11 if ddp else 2):
                wl.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for prm in wl.raw.parameters():                      # stale packs must be re-recorded inside the capture
            getattr(prm, "_osa_packs", {}).clear()
        wl.opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        # under DDP the RCCL watchdog thread keeps polling its work events (hipEventQuery) while this thread captures: in the default
        # "global" capture mode that call from ANOTHER thread invalidates the capture / aborts the process (seen once in r5:
        # ProcessGroupNCCL::Watchdog -> finishedGPUExecutionInternal -> HIP error); "thread_local" confines the checks to this thread
        with torch.cuda.graph(graph, capture_error_mode="thread_local" if ddp else "global"):
            graph_out = wl.step()

        def step():
            graph.replay()
            return graph_out
        step()
        torch.cuda.synchronize()
        assert torch.isfinite(graph_out).all()
        return graph, step
    except Exception as ex:
        print(f"[bench] hipGraph capture of the training step failed ({type(ex).__name__}: {str(ex)[:300]}); running eagerly", file=sys.stderr)
        wl.static = False
        torch.cuda.synchronize()
        return None


def capture_inference_step(eager_step):
    """The inference forwards are fixed sequences of launches on static buffers: capture the (already warmed-up) step once into a
    hipGraph.  Returns (graph, step); (None, eager_step) when capture is unsupported.  The step returns the graph's static output."""
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            graph_out = eager_step()

        def step():
            graph.replay()
            return graph_out
        step()
        torch.cuda.synchronize()
        return graph, step
    except Exception as ex:                      # capture unsupported -> eager launches
        print(f"[bench] hipGraph capture failed ({type(ex).__name__}: {ex}); running eagerly", file=sys.stderr)
        torch.cuda.synchronize()
        return None, eager_step


def gwcnet_timed_config_parity(wl, step, replays=3):
    """Parity of the TIMED configuration itself: the step as it is timed -- all B pairs, `nstreams` concurrent
    sub-batch streams, hipGraph replay, split-format volume -- replayed `replays` times and compared, for EVERY pair, with
    (a) the previous replay (a race between sub-batch streams or a stale arena slot shows up as run-to-run differences),
    (b) the same sub-batches launched eagerly one after the other on ONE stream (same arithmetic, same tiles: bit for bit),
    (c) B single-pair single-stream forwards (f16x3: the per-tensor power-of-two operand scales follow the batch, so agreement is to
        ~1e-6 relative, not bitwise; exact-f32 mode: bitwise).
    Returns the maxima over pairs; tests/test_gpu_timed_config.py asserts on them, bench.py prints them in `config.timed_config_parity`."""
    outs = []
    for _ in range(replays):
        outs.append(step().clone())
    torch.cuda.synchronize()
    per = wl.B // wl.nstreams
    with torch.no_grad():
        seq = torch.cat([wl.net({"left": wl.L[i:i + per], "right": wl.R[i:i + per]})["disp_pred"] for i in range(0, wl.B, per)], 0)
        one = torch.cat([wl.net({"left": wl.L[i:i + 1], "right": wl.R[i:i + 1]})["disp_pred"] for i in range(wl.B)], 0)
    torch.cuda.synchronize()
    out = outs[-1]
    d1 = (out - one).abs().flatten(1)
    if os.environ.get("OSA_PARITY_DIAG"):            # which pairs / how many pixels differ between replays and against the one-stream run
        for r, o in enumerate(outs[1:], 1):
            d = (o - outs[0]).abs().flatten(1)
            print(f"[parity diag] replay {r} vs 0: per-pair max {[round(float(v), 4) for v in d.max(1).values]}, pixels > 1e-3: {[int(v) for v in (d > 1e-3).sum(1)]}", file=sys.stderr)
        d = (out - seq).abs().flatten(1)
        print(f"[parity diag] last replay vs one-stream sub-batches: per-pair max {[round(float(v), 4) for v in d.max(1).values]}, pixels > 1e-3: {[int(v) for v in (d > 1e-3).sum(1)]}", file=sys.stderr)
        from openstereo_amd import ranges
        print(f"[parity diag] sub-batch streams {[hex(s.cuda_stream) for s in wl.sub.streams]}, current {hex(torch.cuda.current_stream().cuda_stream)}, "
              f"arenas {[(hex(k[1]), a[1], a[2]) for k, a in ranges._arenas.items()]}", file=sys.stderr)
    return {"pairs": wl.B, "streams": wl.nstreams, "replays": replays,
            "replay_vs_replay_max_px": max([float((o - outs[0]).abs().max()) for o in outs[1:]] + [0.0]),
            "vs_same_sub_batches_on_one_stream_max_px": float((out - seq).abs().max()),
            "vs_single_pair_runs_max_px": float(d1.max()), "vs_single_pair_runs_worst_pair_epe_px": float(d1.mean(1).max()),
            "all_finite": bool(torch.isfinite(out).all()), "disp_std_min_over_pairs": float(out.flatten(1).std(1).min())}


def _time_steps(step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def eager_training_baseline(name, args, dev, rank, steps=3, warmup=2):
    """`pytorch_rocm_eager_same_gpu` for a training workload: the same class, optimizer, crop and batch, with every hot-path op executed by
    stock PyTorch-ROCm (see _eager_torch_mode).  MIOpen's find mode runs during the warm-up steps."""
    try:
        with _eager_torch_mode():
            wl = WORKLOADS[name](args, dev, rank)
            sec, _ = _time_steps(wl.step, steps, warmup)
        del wl
        torch.cuda.empty_cache()
        return {"value": round((args.batch or 1) / sec, 3), "unit": "stereo-pairs/s", "ms_per_step": round(sec * 1e3, 2),
                "what": f"same workload class / optimizer / crop with the hot-path ops run by PyTorch-ROCm eager (MIOpen), {steps} steps after {warmup} warm-ups"}
    except Exception as ex:
        print(f"[bench] eager training baseline skipped ({type(ex).__name__}: {ex})", file=sys.stderr)
        return None


def _amp_step(step):
    """the step inside an fp16 autocast region"""
    def f():
        with torch.autocast("cuda", dtype=torch.float16):
            return step()
    return f


def secondary_workloads(args, dev, rank, budget_s=170.0):
    """Compact measurements of the other BASELINE configs after the headline: the driver's default line then carries
    LightStereo KITTI15 (configs[3]), the IGEV x32 loop (configs[4]), StereoBase whole-model inference and the StereoBase training steps
    (configs[2]) -- value, ms/step and the dominant launch's roofline fraction each; the time budget bounds the extra run time."""
    out, t_start = {}, time.perf_counter()
    # (name, steps, warm-ups, amp): amp = the same workload inside torch.autocast(fp16), the way the reference runs its AMP configs
    # (cfgs/lightstereo/*: AMP true, cfgs/igev/igev_sceneflow_amp.yaml, cfgs/stereobase/stereobase_sceneflow.yaml:50;
    # trainer_template.py:281) -- the engine layers then use the native f16 mode (engine.effective_precision)
    plan = [("lightstereo_kitti15", 10, 3, False), ("lightstereo_kitti15", 10, 3, True), ("igev_refine32", 5, 2, False), ("igev_refine32", 5, 2, True),
            ("stereobase_e2e", 5, 2, False), ("stereobase_e2e", 5, 2, True), ("stereobase_train", 10, 3, False), ("stereobase_train", 10, 3, True),
            ("stereobase_e2e_train", 3, 2, False), ("stereobase_e2e_train", 3, 2, True)]
    for wname, steps, warmup, amp in plan:
        name = wname + ("_amp" if amp else "")
        if time.perf_counter() - t_start > budget_s:
            out[name] = {"skipped": "time budget of the default run exhausted"}
            continue
        try:
            a = argparse.Namespace(**{**vars(args), "batch": None, "workload": wname, "amp": amp})
            wl = WORKLOADS[wname](a, dev, rank)
            if amp and not wl.training:          # (training workloads run the reference's AMP step themselves: autocast + GradScaler, _amp_training_step)
                wl.step = _amp_step(wl.step)
            step = wl.step
            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            launch = "eager"
            # the same step launched kernel by kernel from Python (what a user of the reference's own eval / train loop gets: tools/measure.py:49-89,
            # trainer_template.py:283 do not capture graphs) -- printed next to the graph-replay figure
            e_sec, _ = _time_steps(step, min(steps, 3), 0)
            if wl.training and not args.no_graph:
                cap = capture_training_step(wl)
                if cap is not None:
                    step, launch = cap[1], "hipGraph replay of the whole training step"
            if wl.graphable and not args.no_graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    gout = wl.step()
                step = lambda g=g, gout=gout: (g.replay(), gout)[1]
                launch = "hipGraph replay"
            sec, res = _time_steps(step, steps, 1)
            assert torch.isfinite(res).all()
            roofs, _ = generic_rooflines(wl, argparse.Namespace(**{**vars(a), "precision": "f16"}) if amp else a, wl.step, 2)
            r0 = roofs[0] if roofs else None
            out[name] = {"metric": wl.metric, "dtype": DTYPES["f16"] if amp else DTYPES[args.precision],
                         "value": round(wl.B / sec, 3), "unit": "stereo-pairs/s", "ms_per_step": round(sec * 1e3, 3),
                         "pairs_per_step": wl.B, "launch": launch,
                         "eager_value": round(wl.B / e_sec, 3), "eager_ms_per_step": round(e_sec * 1e3, 3),
                         "dominant_launch": None if r0 is None else {k: r0[k] for k in ("what", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")}}
            del wl
            torch.cuda.empty_cache()
            if out[name].get("launch", "").startswith("hipGraph replay of the whole training") or (wname.endswith("_train") and not args.no_cpu_baseline):
                # the same training workload with every hot-path op run by stock PyTorch-ROCm (MIOpen) on this GPU, printed beside the engine's
                # figure so that the ratio is driver-visible (VERDICT r3 #3d)
                if time.perf_counter() - t_start < budget_s:
                    eb = eager_training_baseline(wname, a, dev, rank, steps=2, warmup=2)
                    out[name]["pytorch_rocm_eager_same_gpu"] = eb
                    if eb:
                        out[name]["speedup_vs_pytorch_rocm_eager"] = round(out[name]["value"] / eb["value"], 2)
        except Exception as ex:
            out[name] = {"error": f"{type(ex).__name__}: {ex}"}
    return out


# ============================================================================================ CPU baseline (GwcNet)
def gwcnet_cpu_baseline(wl, gpu_out):
    """SURVEY 8d procedure: the reference modules through the import shim when the checkout is mounted, else the oracle
    restatement; fp32, no_grad, all host cores, 1 warm-up + 3 timed full-size pairs, median, per-stage split."""
    from oracle import torch_ref as O           # CPU-baseline leg only (the checker / the "port")
    Lc, Rc, sd = wl.L[:1].cpu(), wl.R[:1].cpu(), wl.sd
    ref_root = os.environ.get("OPENSTEREO_REF", "/root/reference")
    kind, run = "port", None
    if os.path.isdir(os.path.join(ref_root, "stereo")):
        try:
            import importlib
            from openstereo_amd import attach
            attach.stub_reference_packages(ref_root)
            RefGwc = importlib.import_module("stereo.modeling.models.gwcnet.gwcnet").GwcNet
            net = RefGwc(_CfgGet(MAX_DISP=192, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=12, DOWNSAMPLE=4, NUM_GROUPS=40))
            net.load_state_dict(sd)
            net.eval()

            def run(stages):
                inputs = {"left": Lc, "right": Rc}
                t = time.perf_counter(); inputs.update(net.Backbone(inputs)); stages["backbone"] = time.perf_counter() - t
                t = time.perf_counter(); inputs.update(net.CostProcessor(inputs)); stages["volume"] = time.perf_counter() - t
                t = time.perf_counter(); out = net.DispProcessor(inputs); stages["aggregation+head"] = time.perf_counter() - t
                return out["inference_disp"]["disp_est"]
            kind = "reference"
        except Exception as ex:
            print(f"[bench] reference not importable ({type(ex).__name__}: {ex}); CPU baseline = oracle port", file=sys.stderr)
    if run is None:
        def run(stages):
            t = time.perf_counter(); lg, lc = O.gwc_features(Lc, sd); rg, rc = O.gwc_features(Rc, sd); stages["backbone"] = time.perf_counter() - t
            t = time.perf_counter()
            vol = torch.cat((O.gwc_volume(lg, rg, 48, 40), O.concat_volume(lc, rc, 48)), 1); stages["volume"] = time.perf_counter() - t
            t = time.perf_counter(); c3 = O.gwc_aggregate(vol, sd); stages["aggregation"] = time.perf_counter() - t
            t = time.perf_counter(); d = O.upsample_regression(c3, 192, Lc.shape[2], Lc.shape[3]); stages["upsample+softargmin"] = time.perf_counter() - t
            return d
    with torch.no_grad():
        O.gwcnet_forward(Lc[..., :64, :128].contiguous(), Rc[..., :64, :128].contiguous(), sd)     # warm-up (thread pool, allocator): small pair
        times, splits, ref = [], [], None
        for _ in range(3):
            st = {}
            t0 = time.perf_counter()
            ref = run(st)
            times.append(time.perf_counter() - t0)
            splits.append(st)
    med = statistics.median(times)
    mid = splits[times.index(med)]
    # Same restatement, same weights, executed by PyTorch-ROCm eager (MIOpen / rocBLAS kernels) on THIS GPU: what the reference's own
    # PyTorch code gets from an MI355X without the engine.  A baseline like the CPU number, never the thing measured or shipped.
    rocm = None
    try:
        dev = gpu_out.device
        Lg, Rg = wl.L[:1].to(dev), wl.R[:1].to(dev)
        sdg = {k: v.to(dev) for k, v in sd.items()}
        with torch.no_grad():
            for _ in range(2):
                dg = O.gwcnet_forward(Lg, Rg, sdg)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); dg = O.gwcnet_forward(Lg, Rg, sdg); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        m2 = statistics.median(ts)
        rocm = {"value": round(1.0 / m2, 3), "unit": "stereo-pairs/s", "ms_per_pair": round(m2 * 1e3, 2),
                "what": "oracle restatement (plain torch ops, fp32) run by PyTorch-ROCm eager on this GPU, 1 pair per step, median of 3 after 2 warm-ups",
                "epe_vs_engine_px": float((gpu_out[:1] - dg).abs().mean())}
        del sdg, dg
        torch.cuda.empty_cache()
    except Exception as ex:
        print(f"[bench] PyTorch-ROCm eager baseline skipped ({type(ex).__name__}: {ex})", file=sys.stderr)
    return {"value": round(1.0 / med, 5), "unit": "stereo-pairs/s", "cores": torch.get_num_threads(), "kind": kind, "pytorch_rocm_eager_same_gpu": rocm,
            "sample": "3 timed full-size 544x960 D=192 GwcNet forwards after a warm-up, fp32, torch.no_grad; value = 1 / median seconds",
            "seconds": [round(t, 3) for t in times], "stage_seconds": {k: round(v, 3) for k, v in mid.items()},
            "epe_gpu_vs_cpu_px": float((gpu_out[:1].cpu() - ref).abs().mean())}


class _CfgGet(dict):
    __getattr__ = dict.__getitem__


# ============================================================================================ main
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one process per GPU over RCCL, exactly what the driver's torch.distributed.run line does
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    # stdout carries ONE JSON line.  Libraries below PyTorch write diagnostics straight to fd 1 (composable_kernel's "GridwiseOp: Problemsize
    # descriptor dimension check failure" while MIOpen probes solvers for the torch-side fp16 convolutions of the *_amp workloads: 240 lines
    # in front of the JSON in r5's first final pass), so fd 1 points at stderr while the bench runs and the line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    pinned = _pin_host_threads(local, world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or plain `python bench.py --gpus N`)"
    dist = None
    if args.stub:
        dev = torch.device("cpu")
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
    else:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        training = getattr(WORKLOADS.get(args.workload), "training", False)
        if args.force_ddp and world == 1 and training:
            os.environ["OSA_BENCH_FORCE_DDP"] = "1"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if training and _use_ddp() and not args.no_graph:
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")     # DDP + graph capture (capture_training_step)
        if world > 1 or os.environ.get("OSA_BENCH_FORCE_DDP") == "1":
            import torch.distributed as dist
            own_port = world == 1 and "RANK" not in os.environ          # --force-ddp picked MASTER_PORT itself: another process may grab it first
            for attempt in range(4):
                try:
                    dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)       # "nccl" == RCCL on ROCm
                    break
                except Exception as ex:                                  # (DistNetworkError EADDRINUSE between _free_port() and the bind)
                    if not own_port or attempt == 3 or "EADDRINUSE" not in str(ex):
                        raise
                    os.environ["MASTER_PORT"] = str(_free_port())
        from openstereo_amd import _lib, engine
        _lib.load()
        engine.set_precision(args.precision)
    from openstereo_amd.parallel import reduce_step_time, whole_job_rate

    ddp_graph = (not args.stub) and dev.type == "cuda" and _use_ddp() and getattr(WORKLOADS.get(args.workload), "training", False) and not args.no_graph
    if ddp_graph:
        # DDP wrappers that will be captured into a hipGraph are constructed on a side stream (PyTorch's DDP + CUDA-graphs recipe)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            wl = WORKLOADS[args.workload](args, dev, rank)
        torch.cuda.current_stream().wait_stream(side)
    else:
        wl = (_Stub if args.stub else WORKLOADS[args.workload])(args, dev, rank)
    B = wl.B
    if args.amp and not wl.training and not args.stub:
        wl.step = _amp_step(wl.step)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    step = eager_step = wl.step
    for _ in range(args.warmup):
        step()
    sync()
    # The inference forwards are fixed sequences of launches on static buffers: capture once into a hipGraph (launch-bound
    # inner loop -> one graph launch per step).  Warm-up has packed every weight, so only kernels (and the caching allocator's
    # graph pool) are recorded.
    graph = None
    if wl.training and not args.no_graph and dev.type == "cuda":
        cap = capture_training_step(wl, ddp=ddp_graph)
        if dist is not None and world > 1:
            # every rank replays or every rank launches eagerly: a mixed job would dead-lock in the first all-reduce
            ok = torch.tensor([1.0 if cap is not None else 0.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok) == 0.0 and cap is not None:
                print("[bench] another rank could not capture its step: running eagerly", file=sys.stderr)
                cap = None
                wl.static = False
        if cap is not None:
            graph, step = cap
            step()
            sync()
    if wl.graphable and not args.no_graph and dev.type == "cuda":
        graph, step = capture_inference_step(eager_step)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    dt = reduce_step_time(time.perf_counter() - t0, dev)       # MAX over ranks
    assert torch.isfinite(out).all()
    rate = whole_job_rate(B, args.steps, world, dt)

    line = {"metric": wl.metric, "value": round(rate, 3), "unit": "stereo-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": wl.scaling,
            "vs_baseline": None, "dtype": "n/a (stub)" if args.stub else DTYPES["f16" if args.amp else args.precision], "data": "synthetic"}
    cfg = wl.config(args)
    if pinned is not None:
        cfg["host_cpus_of_rank0"] = f"{pinned[0]}-{pinned[-1]} ({len(pinned)} of the job's CPUs, {torch.get_num_threads()} intra-op threads)"
    cfg.update({"pairs_per_gpu_per_step": B, "parallelism": (f"DDP x{world} (RCCL all-reduce of gradients)" if wl.training else f"independent pairs x{world}"),
                "precision": args.precision, "launch": ("hipGraph replay" + (" of the whole training step (forward + loss + backward + optimizer)" if wl.training else ""))
                if graph is not None else "eager"})
    line["config"] = cfg

    if args.workload == "gwcnet" and not args.stub and rank == 0:
        from openstereo_amd import engine
        roofs, alt, latency_1, cpu = [], None, None, None
        if not args.timed_only and world == 1:
            cfg["timed_config_parity"] = gwcnet_timed_config_parity(wl, step)      # every pair of the timed step vs single-pair single-stream runs
            # the same maxima as scalar keys: the driver's parsed copy of `config` keeps scalars only
            tp = cfg["timed_config_parity"]
            cfg.update({"parity_replay_vs_replay_max_px": tp["replay_vs_replay_max_px"], "parity_vs_one_stream_max_px": tp["vs_same_sub_batches_on_one_stream_max_px"],
                        "parity_vs_single_pair_runs_max_px": tp["vs_single_pair_runs_max_px"], "parity_worst_pair_epe_px": tp["vs_single_pair_runs_worst_pair_epe_px"],
                        "parity_all_finite": tp["all_finite"]})
        if not args.timed_only:
            nrep = max(2, min(args.steps, 5))
            if graph is not None and world == 1:
                # like for like with the reference's loops, which launch eagerly: the same step, same arithmetic mode, no graph
                e_sec, _ = _time_steps(eager_step, nrep, 1)
                line["eager_value"], line["eager_ms_per_step"] = round(B / e_sec, 3), round(e_sec * 1e3, 3)
            roofs, per_step = gwcnet_rooflines(wl, args, wl.step_single, nrep)
            if roofs and wl.nstreams > 1:
                roofs[0]["note"] = (f"kernel timed in a single-stream replay of the same forward ({B} pairs per launch, nothing else on the GPU); the timed region "
                                    f"issues every launch as {wl.nstreams} concurrent launches of {B // wl.nstreams} pairs on separate streams (config.sub_batch_streams)")
            cfg["stage_ms_per_step"] = dict(list(per_step.items())[:14])
            cfg["stage_top6_ms_per_step"] = "; ".join(f"{k}={v}" for k, v in list(per_step.items())[:6])       # (scalar copy for the driver's parsed config)
            if args.stages:
                json.dump(per_step, open(args.stages, "w"), indent=1)
            if world == 1:
                # the other arithmetic mode, same workload (packed weights follow the mode switch by themselves)
                other = "f32" if args.precision == "f16x3" else "f16x3"
                engine.set_precision(other)
                for _ in range(2):
                    eager_step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(nrep):
                    eager_step()
                torch.cuda.synchronize()
                t_other = (time.perf_counter() - t1) / nrep
                args_o = argparse.Namespace(**{**vars(args), "precision": other})
                r_other, _ = gwcnet_rooflines(wl, args_o, wl.step_single, 2)
                alt = {"precision": other, "dtype": DTYPES[other], "value": round(B / t_other, 3), "unit": "stereo-pairs/s",
                       "ms_per_step": round(t_other * 1e3, 3), "launch": "eager",
                       "roofline": None if not r_other else {k: r_other[0][k] for k in ("kernel", "achieved", "peak", "frac", "avg_launch_ms")}}
                engine.set_precision(args.precision)
                out = eager_step()
                # single-pair latency ("report B=1 latency and best-throughput B"), eager launches
                if B != 1:
                    with torch.no_grad():
                        one = {"left": wl.L[:1], "right": wl.R[:1]}
                        for _ in range(3):
                            wl.net(dict(one))
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        for _ in range(10):
                            wl.net(dict(one))
                        torch.cuda.synchronize()
                        latency_1 = round((time.perf_counter() - t1) / 10 * 1e3, 3)
                if not args.no_cpu_baseline:
                    cpu = gwcnet_cpu_baseline(wl, out)
        cfg["latency_ms_1_pair"] = latency_1
        if roofs:
            # `traffic` of every roofline record is the PMC figure of profiles/traffic.json (separate --pmc passes, tools/profile_round4.sh):
            # say which code state it was collected at
            try:
                tm = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("_measured_at")
            except Exception:
                tm = None
            for r in roofs:
                if r.get("traffic") is not None:
                    r["traffic_measured_at"] = tm
        if cpu is not None:
            cfg["cpu_baseline_kind"] = (f"{cpu['kind']}: " + ("the reference's own modules through the import shim" if cpu["kind"] == "reference" else
                                        "the oracle restatement oracle/torch_ref.py (the reference checkout is not mounted on this box)"))
        line["roofline"] = roofs[0] if roofs else None
        line["rooflines"] = roofs[1:]
        line["cpu_baseline"] = cpu
        line["other_precision"] = alt
        if not args.timed_only and world == 1 and not args.no_workloads:
            del wl
            torch.cuda.empty_cache()
            line["workloads"] = secondary_workloads(args, dev, rank)
    elif rank == 0:
        line["roofline"], line["cpu_baseline"] = None, None
        if not args.stub and not args.timed_only and dev.type == "cuda":
            roofs, per_step = generic_rooflines(wl, args, eager_step, max(2, min(args.steps, 3)))
            cfg["stage_ms_per_step"] = dict(list(per_step.items())[:14])
            line["roofline"] = roofs[0] if roofs else None
            line["rooflines"] = roofs[1:]
            line["cpu_baseline_note"] = "the CPU leg runs with the default workload only (bench.py without --workload)"
            if wl.training and world == 1 and not args.no_cpu_baseline:
                del wl
                torch.cuda.empty_cache()
                line["pytorch_rocm_eager_same_gpu"] = eager_training_baseline(args.workload, args, dev, rank)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
