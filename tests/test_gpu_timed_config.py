"""The configuration bench.py TIMES, parity-tested as it is timed (VERDICT r4 weak #2 / next #1c).

bench.py's default step is not the single-pair eager forward the golden tests run: it is 9 pairs per step, split into 3 sub-batches on 3
concurrent HIP streams (parallel.SubBatchStreams: per-stream range arenas, event-ordered weight packs), captured once into a hipGraph and
replayed, with the cost volume written in the f16x3 chain's split format.  This test builds exactly that object through bench.py's own
workload class and capture function, replays it three times and compares EVERY pair with (a) the other replays, (b) the same sub-batches
launched eagerly on one stream, (c) nine single-pair single-stream forwards, and pair 0 with the reference's own full-size golden."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

from conftest import golden, ROOT
from openstereo_amd.utils.weights import synth_images

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_timed_configuration_every_pair(prec):
    from openstereo_amd import engine
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    old = engine.get_precision()
    engine.set_precision(prec)
    try:
        dev = torch.device("cuda", 0)
        if os.environ.get("OSA_TEST_RESET_ARENAS"):
            from openstereo_amd import ranges
            ranges.reset_arenas()
        wl = bench.GwcNetInference(argparse.Namespace(batch=None, streams=None), dev, 0)
        assert wl.B == 9 and wl.nstreams == 3, "bench.py's default line: 9 pairs as 3 sub-batch streams of 3"
        # pair 0 := the input of tests/golden/gwcnet_full_disp.npz (the reference's own forward at 544x960, D=192); same weights (seed 0)
        L0, R0 = synth_images(1, 544, 960, seed=1)
        wl.L[0].copy_(L0[0].to(dev)); wl.R[0].copy_(R0[0].to(dev))
        eager = wl.step
        for _ in range(3):                       # bench.py's default warm-up (first call chained, weights packed, arenas built)
            eager()
        torch.cuda.synchronize()
        graph, step = bench.capture_inference_step(eager)
        assert graph is not None, "hipGraph capture of the timed step failed"
        par = bench.gwcnet_timed_config_parity(wl, step, replays=3)
        print(prec, par)
        assert par["all_finite"] and par["disp_std_min_over_pairs"] > 1.0
        assert par["replay_vs_replay_max_px"] == 0.0, par
        assert par["vs_same_sub_batches_on_one_stream_max_px"] == 0.0, par
        # (c): f16x3 operand scales are per-tensor maxima over the batch and tile choices follow the pixel count of a launch, so the
        # single-pair runs agree to ~1e-6 relative (disparities ~100 px), not bitwise (test_gwcnet_batch_invariance_and_odd_size)
        # [MI355X] r5: f16x3 max 4.3e-4 px, worst-pair EPE 3.8e-5 px (the size of the mode's distance to the fp32 reference, 5e-5 px)
        if prec == "f32":
            assert par["vs_single_pair_runs_max_px"] < 1e-4 and par["vs_single_pair_runs_worst_pair_epe_px"] < 1e-6, par
        else:
            assert par["vs_single_pair_runs_max_px"] < 2e-3 and par["vs_single_pair_runs_worst_pair_epe_px"] < 1e-4, par
        out = step()
        torch.cuda.synchronize()
        g = golden("gwcnet_full_disp.npz")
        epe = float(np.abs(out[0].cpu().numpy() - g["disp"][0]).mean())
        assert epe < 1e-3, epe
    finally:
        engine.set_precision(old)
        engine.MULTI_STREAM = False
