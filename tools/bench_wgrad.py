"""GPU: weight-gradient kernels per layer shape, f16x3 form vs exact-fp32 form (ms per call incl. the reduce stage; TFLOP/s)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openstereo_amd import autograd as AG, ops
from openstereo_amd.ranges import input_meta
SHAPES = [("3d 32->32 @48x64x128", 32, 32, (3, 3, 3), (48, 64, 128)), ("3d 64->32 @48x64x128", 64, 32, (3, 3, 3), (48, 64, 128)),
          ("3d 64->64 @24x32x64", 64, 64, (3, 3, 3), (24, 32, 64)), ("3d 128->128 @12x16x32", 128, 128, (3, 3, 3), (12, 16, 32)),
          ("3d 1x1x1 32->32 @48x64x128", 32, 32, (1, 1, 1), (48, 64, 128)), ("3d 32->1 @48x64x128", 32, 1, (3, 3, 3), (48, 64, 128)),
          ("2d 384->128 @80x184 (gru04)", 384, 128, (1, 3, 3), (1, 80, 184)), ("2d 128->256 @80x184 (head)", 128, 256, (1, 3, 3), (1, 80, 184)),
          ("2d 256->128 @20x46 (gru16)", 256, 128, (1, 3, 3), (1, 20, 46))]
ONLY = [a for a in sys.argv[1:] if a not in ("f16x3", "f32", "f16")]
PRECS = [a for a in sys.argv[1:] if a in ("f16x3", "f32", "f16")] or ["f16x3", "f32"]
NB = int(os.environ.get("BATCH", "1"))          # batch items (22 = the batched gradient of the update block's GRU iterations)
SHAPES += [("2d 384->256 @80x184 (gru08 rz)", 384, 256, (1, 3, 3), (1, 80, 184)), ("2d 64->64 @80x184", 64, 64, (1, 3, 3), (1, 80, 184)),
           ("2d 452->64 k1 @80x184", 452, 64, (1, 1, 1), (1, 80, 184)), ("3d 48->48 @24x40x92", 48, 48, (3, 3, 3), (24, 40, 92))]
for name, Ci, Co, k, (D, H, W) in SHAPES:
    if ONLY and not any(o in name for o in ONLY):
        continue
    x = ops.to_cl(torch.randn(NB, Ci, D, H, W, device="cuda"))
    dy = ops.to_cl(torch.randn(NB, Co, D, H, W, device="cuda") * 1e-3)
    dw = torch.empty(Co, Ci, *k, device="cuda")
    pad = tuple(kk // 2 for kk in k)
    mx, mdy = input_meta(x), input_meta(dy)
    res = {}
    for prec in PRECS:
        call = lambda: AG._wgrad(x, dy, dw, NB, D, H, W, Ci, D, H, W, Co, k, 1, pad, (1, 1, 1), 0, prec, mx, mdy)
        for _ in range(3):
            call()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            call()
        torch.cuda.synchronize(); res[prec] = (time.perf_counter() - t0) / 20
    fl = 2.0 * NB * D * H * W * Ci * Co * k[0] * k[1] * k[2]
    print(f"{name:34s} " + "   ".join(f"{p_} {res[p_] * 1e3:7.3f} ms ({fl / res[p_] / 1e12:6.1f} TF/s)" for p_ in PRECS)
          + (f"   x{res['f32'] / res['f16x3']:.2f}" if ("f32" in res and "f16x3" in res) else ""))
