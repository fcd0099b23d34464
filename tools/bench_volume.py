"""Micro-benchmark of the NDHWC volume builder (GwcNet shape): quad-lane kernel vs the per-channel
kernel and timing-only ablations (OSA_VOL_DBG).  GPU only.
"""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from openstereo_amd import ops
dev = "cuda:0"
B, H, W, D = int(os.environ.get("VOL_B", 1)), 136, 240, 48
gf = ops.empty_cl(2 * B, 320, 1, H, W, dev); gf.normal_()
cf = ops.empty_cl(2 * B, 12, 1, H, W, dev); cf.normal_()
from openstereo_amd import ranges
ranges.ensure_meta(gf); ranges.ensure_meta(cf)          # (the split form derives its scale from the features' range blocks)
SPLIT = False
def run():
    return ops.build_cost_volume_from_cl(gf, 40, cf, B, D, out_split=SPLIT)
outs = {}
from openstereo_amd import _lib
LIB = _lib.load()
MODES = os.environ.get("VOL_MODES", "quads,perchannel,quads,px2,quads,px2,w4,w8,w8lds160,dbg1,dbg2,dbg4,dbg7,px2dbg1,px2dbg7").split(",")
for mode in MODES:
    if hasattr(LIB, "osa_volume_walk_step"):      # walk8 / walk4: the d-walking form (r4); every other mode measures the chunked kernel
        LIB.osa_volume_walk_step(int(mode[4:5]) if mode.startswith("walk") else 0)
    SPLIT = mode.endswith("split")                      # walk8split: the volume written as a split tensor (f16x3 chains)
    for k in ("OSA_VOL_PERCHANNEL", "OSA_VOL_DBG", "OSA_VOL_WAVES", "OSA_VOL_LDS", "OSA_VOL_PX2"):
        os.environ.pop(k, None)
    if mode.startswith("w") and not mode.startswith("px2"):
        os.environ["OSA_VOL_WAVES"] = mode[1]
        if "lds" in mode: os.environ["OSA_VOL_LDS"] = str(int(mode.split("lds")[1]) * 1024)
    if mode == "perchannel": os.environ["OSA_VOL_PERCHANNEL"] = "1"
    if mode.startswith("px2"): os.environ["OSA_VOL_PX2"] = "1"
    if "dbg" in mode: os.environ["OSA_VOL_DBG"] = mode.split("dbg")[1]
    for _ in range(3): v = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): v = run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    outs[mode] = v.clone()
    print(f"{mode}: {ms:.3f} ms  {(487.8e6 * B / ms / 1e9):.2f} TB/s (algorithmic 487.8 MB)")
for a in ("perchannel", "px2", "walk8", "walk4"):
    if a in outs and "quads" in outs:
        print(f"quads vs {a} bit-identical:", torch.equal(outs["quads"], outs[a]))
