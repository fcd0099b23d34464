"""Which launches still marshal through ctypes when the C++ extension is loaded?  One training step (after a warm-up step, so one-time
weight packing is listed separately) and one inference forward of the small StereoBase / GwcNet / IGEV / LightStereo test models.
    python tools/ctypes_census.py"""
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
from openstereo_amd import _ext, _lib                                        # noqa: E402
from openstereo_amd.utils.weights import synth_state_dict, synth_images      # noqa: E402

DEV = "cuda"


def census(tag, fn, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    first = dict(_lib.CALLS)
    _lib.CALLS.clear()
    fn()
    torch.cuda.synchronize()
    steady = dict(_lib.CALLS)
    _lib.CALLS.clear()
    print(f"[{tag}] ctypes calls in a steady-state call: {sum(steady.values())} {dict(sorted(steady.items(), key=lambda t: -t[1]))}")
    once = {k: v for k, v in first.items() if k not in steady}
    if once:
        print(f"[{tag}]   first call only: {once}")


def main():
    print("extension loaded:", _ext.load() is not None)
    from openstereo_amd.models import stereo_models as SM
    from openstereo_amd.models.gwcnet import GwcNet
    L, R = synth_images(1, 64, 128, seed=1)
    L, R = L.to(DEV), R.to(DEV)
    L2, R2 = synth_images(1, 128, 256, seed=31, max_shift=12.0)
    L2, R2 = L2.to(DEV), R2.to(DEV)
    L255, R255 = (L2 * 40 + 128).clamp(0, 255), (R2 * 40 + 128).clamp(0, 255)

    def train_step(net, l, r):
        def f():
            net.zero_grad(set_to_none=True)
            out = net({"left": l, "right": r})
            (sum(p.float().abs().mean() for p in out["disp_preds"]) + (out["init_disp"].abs().mean() if "init_disp" in out else 0.0)).backward()
        return f

    def infer(net, l, r):
        def f():
            with torch.no_grad():
                net({"left": l, "right": r})
        return f
    g = GwcNet()
    g.load_state_dict(synth_state_dict(g, seed=0))
    g = g.to(DEV)
    census("GwcNet train", train_step(g.train(), L, R))
    census("GwcNet eval", infer(g.eval(), L, R))
    cfg = SimpleNamespace(MAX_DISP=64, NUM_GROUPS=8, USE_CONCAT_VOLUME=True, CONCAT_CHANNELS=8, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3,
                          CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=False, EVAL_ITERS=4, TRAIN_ITERS=4)
    s = SM.StereoBase(cfg)
    s.load_state_dict(synth_state_dict(s, seed=41, head_gain=20.0, gain=0.9))
    s = s.to(DEV)
    census("StereoBase train", train_step(s.train(), L2, R2))
    with torch.autocast("cuda", dtype=torch.float16):
        census("StereoBase train, autocast", train_step(s.train(), L2, R2))
    census("StereoBase eval", infer(s.eval(), L2, R2))
    args = SimpleNamespace(MAX_DISP=64, HIDDEN_DIMS=[128, 128, 128], N_GRU_LAYERS=3, CORR_RADIUS=4, CORR_LEVELS=2, SLOW_FAST_GRU=True, VALID_ITERS=4,
                           TRAIN_ITERS=4, N_DOWNSAMPLE=2)
    i = SM.IGEVStereo(args)
    i.load_state_dict(synth_state_dict(i, seed=43))
    i = i.to(DEV)
    census("IGEV train", train_step(i.train(), L255, R255))
    census("IGEV eval", infer(i.eval(), L255, R255))
    lcfg = SimpleNamespace(MAX_DISP=192, LEFT_ATT=True, AGGREGATION_BLOCKS=[1, 2, 4], EXPANSE_RATIO=4)
    ls = SM.LightStereo(lcfg)
    ls.load_state_dict(synth_state_dict(ls, seed=47))
    ls = ls.to(DEV)
    census("LightStereo eval", infer(ls.eval(), L2, R2))


if __name__ == "__main__":
    main()
