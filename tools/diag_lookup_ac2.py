import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openstereo_amd.geometry import _Lookup
torch.manual_seed(0)
def P(*a): print(*a, flush=True)
g0 = torch.randn(1, 16, 32, 24, 16, device="cuda"); g1 = torch.randn(1, 16, 32, 24, 8, device="cuda")
c0 = torch.randn(1, 16, 32, 32, device="cuda"); c1 = torch.randn(1, 16, 32, 16, device="cuda")
d = torch.rand(1, 16, 32, device="cuda") * 10; cx = torch.arange(32, device="cuda").float().reshape(1, 1, 32).repeat(1, 16, 1).contiguous()
ref = _Lookup.apply(d, cx, 24, 4, g0, g1, c0, c1); torch.cuda.synchronize()
refh = _Lookup.apply(d, cx, 24, 4, g0, g1, c0.half().float(), c1.half().float()); torch.cuda.synchronize()
P("fp16-rounded corr, no autocast: err vs ref", float((refh - ref).abs().max()))
def stats(tag, o):
    e = (o - refh).abs()
    per = e.reshape(1, 2, 25 * 9, 16, 32).amax(dim=(0, 3, 4))      # [level, (C+1)*taps]
    P(tag, "max err", float(e.max()), " geo part L0 %.3g corr part L0 %.3g | geo L1 %.3g corr L1 %.3g" % (
        float(per[0, :24 * 9].max()), float(per[0, 24 * 9:].max()), float(per[1, :24 * 9].max()), float(per[1, 24 * 9:].max())))
with torch.autocast("cuda", dtype=torch.float16):
    o = _Lookup.apply(d, cx, 24, 4, g0, g1, c0.half().float(), c1.half().float()); torch.cuda.synchronize(); stats("A all fp32 under autocast:", o)
    o = _Lookup.apply(d, cx, 24, 4, g0, g1, c0.half(), c1.half()); torch.cuda.synchronize(); stats("B fp16 corr under autocast:", o)
    o = _Lookup.apply(d, cx, 24, 4, g0.half(), g1.half(), c0.half().float(), c1.half().float()); torch.cuda.synchronize()
    P("D fp16 geo under autocast: max", float(o.abs().max()))
    t0, t1 = c0.half(), c1.half()
    o = _Lookup.apply(d, cx, 24, 4, g0, g1, t0, t1); torch.cuda.synchronize(); stats("B2 fp16 corr kept alive:", o)
    o = _Lookup.apply(d.half(), cx, 24, 4, g0, g1, c0.half().float(), c1.half().float()); torch.cuda.synchronize(); stats("E fp16 disp:", o)
