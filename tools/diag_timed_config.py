"""Replay-to-replay determinism of bench.py's timed GwcNet configuration (9 pairs, sub-batch streams, hipGraph): r5 diagnosis.
    python tools/diag_timed_config.py [--streams N] [--no-graph] [--replays K] [--batch B] [--warm W]
Environment switches of the library / Python layer apply (OSA_LIB_PATH, OSA_B_RING_MASK, OSA_VOL_WALK, OSA_VOL_SPLIT, OSA_SPLIT_ACT, ...)."""
import argparse
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=None)
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--replays", type=int, default=6)
ap.add_argument("--warm", type=int, default=3)
ap.add_argument("--precision", default="f16x3")
ap.add_argument("--tag", default="")
ap.add_argument("--stages", action="store_true", help="eager multi-stream steps with every stage of the 3-D aggregation stashed: which stage differs first from the one-stream run?")
a = ap.parse_args()
from openstereo_amd import engine, _lib  # noqa: E402
_lib.load()
engine.set_precision(a.precision)
dev = torch.device("cuda", 0)
wl = bench.GwcNetInference(argparse.Namespace(batch=a.batch, streams=a.streams), dev, 0)
for _ in range(a.warm):
    wl.step()
torch.cuda.synchronize()
step = wl.step
if not a.no_graph:
    g, step = bench.capture_inference_step(wl.step)
    assert g is not None
if a.stages:
    import openstereo_amd.models.gwcnet as G
    per = wl.B // wl.nstreams

    def run(fn):
        G.STAGE_STASH = []
        o = fn()
        torch.cuda.synchronize()
        st, G.STAGE_STASH = G.STAGE_STASH, None
        return o, st
    with torch.no_grad():
        ref = []
        for i in range(0, wl.B, per):            # one stream, the same sub-batches one after the other
            ref.append(run(lambda: wl.net({"left": wl.L[i:i + per], "right": wl.R[i:i + per]})["disp_pred"]))
        ref2 = run(lambda: wl.net({"left": wl.L[0:per], "right": wl.R[0:per]})["disp_pred"])
    same = all(torch.equal(x[1].view(torch.int32), y[1].view(torch.int32)) for x, y in zip(ref[0][1], ref2[1]))
    print(f"one-stream run repeated: every stage bit-identical: {same}")
    names = [n for n, _ in ref[0][1]]
    nst = len(names)
    for r in range(a.replays):
        o, st = run(wl.step)                     # the sub-batches interleave in the stash: branch i's stages are the i-th occurrence of each name
        seen = {}
        for n, t in st:
            i = seen.get(n, 0)
            seen[n] = i + 1
            want = dict(ref[i][1])[n]
            neq = (t.view(torch.int32) != want.view(torch.int32))
            c = int(neq.sum())
            if c:
                idx = neq.nonzero()[0].tolist()
                print(f"  step {r} sub-batch {i} stage {n:16s} shape {tuple(t.shape)}: {c} differing 32-bit words, first at {idx}")
        d = (o - torch.cat([x[0] for x in ref], 0)).abs().flatten(1)
        print(f"  step {r}: disparity pixels > 1e-3 per pair {[int(v) for v in (d > 1e-3).sum(1)]}")
    sys.exit(0)
outs = []
for _ in range(a.replays):
    outs.append(step().clone())
torch.cuda.synchronize()
per = wl.B // wl.nstreams
with torch.no_grad():
    seq = torch.cat([wl.net({"left": wl.L[i:i + per], "right": wl.R[i:i + per]})["disp_pred"] for i in range(0, wl.B, per)], 0)
torch.cuda.synchronize()
bad = 0
for r, o in enumerate(outs):
    d = (o - seq).abs().flatten(1)
    n = [int(v) for v in (d > 1e-3).sum(1)]
    bad += sum(n)
    if sum(n):
        print(f"  replay {r} vs one-stream eager: pixels > 1e-3 per pair {n}, max {float(d.max()):.3f}")
        b = int(torch.tensor(n).argmax())
        idx = ((o[b] - seq[b]).abs() > 1e-3).nonzero()
        ys, xs = idx[:, 0], idx[:, 1]
        print(f"    pair {b}: rows {int(ys.min())}..{int(ys.max())}, cols {int(xs.min())}..{int(xs.max())}; first pixels {idx[:6].tolist()}")
print(f"[{a.tag or 'default'}] streams={wl.nstreams} graph={not a.no_graph} precision={a.precision}: {bad} differing pixels over {a.replays} replays"
      + ("" if bad else "  -- deterministic"))
