import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openstereo_amd.geometry import CombinedGeoEncodingVolume
torch.manual_seed(0)
B, C, D, H, W = 1, 24, 16, 16, 32
f1 = torch.randn(B, 96, H, W, device="cuda").requires_grad_()
f2 = torch.randn(B, 96, H, W, device="cuda").requires_grad_()
gv = torch.randn(B, C, D, H, W, device="cuda").requires_grad_()
disp = torch.rand(B, 1, H, W, device="cuda") * 10
coords = torch.arange(W, device="cuda").float().reshape(1, 1, W, 1).repeat(B, H, 1, 1)
ref = CombinedGeoEncodingVolume(f1, f2, gv)(disp, coords)
print("fp32 path: max", float(ref.abs().max()))
for low in (False, True):
    with torch.autocast("cuda", dtype=torch.float16):
        a, b, g = (f1.half(), f2.half(), gv.half()) if low else (f1, f2, gv)
        fn = CombinedGeoEncodingVolume(a.float(), b.float(), g.float())
        print("  pyramid dtypes", [t.dtype for t in fn.geo_volume_pyramid], [t.dtype for t in fn.init_corr_pyramid], [t.is_contiguous() for t in fn.init_corr_pyramid])
        out = fn(disp, coords)
    torch.cuda.synchronize()
    print("autocast low=%s: dtype %s max %.4g  err vs fp32 %.4g" % (low, out.dtype, float(out.abs().max()), float((out - ref).abs().max())))
    out.float().sum().backward()
    print("   grads finite:", bool(torch.isfinite(f1.grad).all()), bool(torch.isfinite(gv.grad).all()))
# hypothesis check, LAST (may fault): F.avg_pool1d on fp16 rows, as the previous train path did under autocast
import torch.nn.functional as F
c = torch.randn(1, 16, 32, 32, device="cuda").half()
print("avg_pool1d fp16 ...", flush=True)
o = F.avg_pool1d(c.reshape(-1, 1, 32), 2, 2); torch.cuda.synchronize()
want = F.avg_pool1d(c.float().reshape(-1, 1, 32), 2, 2)
print("avg_pool1d fp16 err", float((o.float() - want).abs().max()), flush=True)
from openstereo_amd.geometry import _Lookup
g0 = torch.randn(1, 16, 32, 24, 16, device="cuda"); g1 = torch.randn(1, 16, 32, 24, 8, device="cuda")
c0 = torch.randn(1, 16, 32, 32, device="cuda"); c1 = torch.randn(1, 16, 32, 16, device="cuda")
d = torch.rand(1, 16, 32, device="cuda") * 10; cx = torch.arange(32, device="cuda").float().reshape(1, 1, 32).repeat(1, 16, 1).contiguous()
ref = _Lookup.apply(d, cx, 24, 4, g0, g1, c0, c1); torch.cuda.synchronize()
print("_Lookup fp32 ok", flush=True)
with torch.autocast("cuda", dtype=torch.float16):
    o = _Lookup.apply(d, cx, 24, 4, g0, g1, c0.half(), c1.half()); torch.cuda.synchronize()
print("_Lookup with fp16 corr levels under autocast: err", float((o - ref).abs().max()), flush=True)
