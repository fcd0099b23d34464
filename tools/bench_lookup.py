"""The training path's geometry-encoding lookup (osa_geo_lookup_f32, NCHW output) and its gradient w.r.t. the pyramid levels
(osa_geo_lookup_bwd_f32) at the StereoBase training map (320x736 crop: 80 x 184 at 1/4, C = 8, D = 48, two levels, radius 4): the forms of
both kernels timed and compared bit for bit.
    bash tools/build_one_variant.sh exp_geo geometry -DOSA_EXPERIMENTS
    OSA_LIB_PATH=openstereo_amd/lib/variants/exp_geo.so python tools/bench_lookup.py
(the form switches exist in the experiments build only: the shipped library always runs the rows forms)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, ".")
from openstereo_amd import _lib     # noqa: E402

dev = "cuda"
B, H, W, C, D, r, L = 1, 80, 184, int(os.environ.get("LOOKUP_C", "8")), 48, 4, 2      # LOOKUP_C: geometry channels (IGEV 8)
g = torch.Generator().manual_seed(0)
disp = (torch.rand(B, H, W, generator=g) * 40).to(dev)
cx = torch.arange(W).float().view(1, 1, W).repeat(B, H, 1).to(dev)
dout = torch.randn(B, (C + 1) * (2 * r + 1) * L, H, W, generator=g).to(dev)
shapes = [(B, H, W, C, D >> l) for l in range(L)] + [(B, H, W, W >> l) for l in range(L)]
st = torch.cuda.current_stream().cuda_stream


def run(form):
    os.environ.pop("OSA_GEO_BWD_SCATTER", None)
    os.environ["OSA_GEO_BWD_FORM"] = "2"
    if form == "scatter":
        os.environ["OSA_GEO_BWD_SCATTER"] = "1"
    elif form == "gather":
        os.environ["OSA_GEO_BWD_FORM"] = "1"
    grads = [torch.full(s, float("nan"), device=dev) for s in shapes]
    gp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in grads[:L]])
    cp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in grads[L:]])
    gl = (ctypes.c_int * L)(*[s[-1] for s in shapes[:L]])
    cl = (ctypes.c_int * L)(*[s[-1] for s in shapes[L:]])
    call = lambda: _lib.call("osa_geo_lookup_bwd_f32", gp, cp, gl, cl, L, disp.data_ptr(), cx.data_ptr(), dout.data_ptr(), B, H, W, C, r, st)
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1000, grads


ref = None
for form in ("scatter", "gather", "rows"):
    us, grads = run(form)
    same = "" if ref is None else f"; bit-identical to the scatter form: {all(torch.equal(a, b) for a, b in zip(grads, ref))}"
    ref = ref or grads
    nbytes = sum(t.numel() for t in grads) * 4 + dout.numel() * 4
    print(f"[lookup bwd, {form:7s}] {us:7.1f} us per call ({nbytes / us / 1e6:.2f} TB/s of written + read bytes){same}")


levels = [torch.randn(s_, generator=g).to(dev) for s_ in shapes]
gpl = (ctypes.c_void_p * L)(*[t.data_ptr() for t in levels[:L]])
cpl = (ctypes.c_void_p * L)(*[t.data_ptr() for t in levels[L:]])
gl_ = (ctypes.c_int * L)(*[s_[-1] for s_ in shapes[:L]])
cl_ = (ctypes.c_int * L)(*[s_[-1] for s_ in shapes[L:]])
ref = None
for form in ("pixel", "rows"):
    os.environ["OSA_GEO_FWD_PIXEL"] = "1" if form == "pixel" else "0"
    out = torch.full_like(dout, float("nan"))
    call = lambda: _lib.call("osa_geo_lookup_f32", gpl, cpl, gl_, cl_, L, disp.data_ptr(), cx.data_ptr(), out.data_ptr(), B, H, W, C, r, st)
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1000
    same = "" if ref is None else f"; bit-identical to the pixel form: {torch.equal(out, ref)}"
    ref = out if ref is None else ref
    print(f"[lookup fwd, {form:7s}] {us:7.1f} us per call{same}")
