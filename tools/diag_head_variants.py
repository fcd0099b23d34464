"""r6: the x4 fused head with chosen classes of its packed-fp32 instructions expanded to scalar code (tools/experiments/head_asm_variants.py),
each variant run next to the loads that make the fully packed build fail; outputs compared bit for bit with the variant's idle-GPU result.

    python tools/experiments/head_asm_variants.py && python tools/diag_head_variants.py [--iters 30] [--loads march,b2,b4,...]

Code objects are loaded with hipModuleLoad and launched with hipModuleLaunchKernel (UpArgs of csrc/softargmin.hip as the kernarg block)."""
import argparse
import ctypes as C
import glob
import os
import struct
import subprocess
import sys

import torch
import torch.nn as nn

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--loads", default="none,march,b2,b4")
ap.add_argument("--variants", default="")
a = ap.parse_args()
from openstereo_amd import _lib, engine, ops  # noqa: E402
from openstereo_amd.engine import PackedConv3d  # noqa: E402

_lib.load()
hip = C.CDLL("libamdhip64.so")
probe = C.CDLL(os.path.join(ROOT, "tools", "experiments", "libpk_probe.so"))
probe.burner_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
KERNEL = b"_ZN3osa27upsample4_softargmin_kernelENS_6UpArgsE"
dev = torch.device("cuda", 0)
try:
    ident = subprocess.run("rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique id:' | head -1", shell=True, capture_output=True, text=True).stdout.strip()
except Exception:
    ident = "?"
print(f"[box] {torch.cuda.get_device_name(0)} | {ident}", flush=True)

g = torch.Generator().manual_seed(1)
B, D, H, W = 3, 48, 136, 240
cost = (torch.randn(B, D, H, W, generator=g) * 3.0).to(dev)
bsrc = torch.randn(1 << 16, generator=g).to(dev)
bsink = torch.zeros(256, device=dev)
side = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()


def load_variant(path):
    mod, fn = C.c_void_p(), C.c_void_p()
    rc = hip.hipModuleLoad(C.byref(mod), path.encode())
    assert rc == 0, (path, rc)
    rc = hip.hipModuleGetFunction(C.byref(fn), mod, KERNEL)
    assert rc == 0, (path, rc)
    return fn


def launcher(fn):
    Ho, Wo = 4 * H, 4 * W
    tiles = B * ((Ho + 3) // 4) * ((Wo + 63) // 64)

    def run():
        out = torch.empty(B, Ho, Wo, device=dev)
        # struct UpArgs { const float* cost; float* out; int B, Dl, Hl, Wl, D, H, W; int align; float sd, sh, sw; }  (64 bytes with tail padding)
        blob = struct.pack("<QQ8i3f4x", cost.data_ptr(), out.data_ptr(), B, D, H, W, 4 * D, Ho, Wo, 0, 0.25, 0.25, 0.25)
        buf = C.create_string_buffer(blob, len(blob))
        size = C.c_size_t(len(blob))
        extra = (C.c_void_p * 5)(1, C.cast(buf, C.c_void_p), 2, C.cast(C.pointer(size), C.c_void_p), 3)
        rc = hip.hipModuleLaunchKernel(fn, tiles, 1, 1, 256, 1, 1, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream), None, extra)
        assert rc == 0, rc
        run.keep = (buf, size, extra)
        return out
    return run


def make_load(name):
    if name == "none":
        return None
    if name == "march":
        conv = nn.Conv3d(32, 32, 3, padding=1, bias=False).to(dev)
        pc0 = PackedConv3d(conv, None, 1, precision="f16x3")
        xs = []
        for _ in range(2):
            t = ops.to_cl(torch.randn(B, 32, D, H, W, generator=g).to(dev))
            t._osa_meta = engine.input_meta(t)
            xs.append(pc0(t, out_split=True))
        torch.cuda.synchronize()

        def queue():
            for st, t in zip(side, xs):
                with torch.cuda.stream(st):
                    for _ in range(3):
                        pc0(t, out_split=True)
        return queue
    kind = int(name[1:])

    def queue():
        for st in side:
            for _ in range(3):
                rc = probe.burner_launch(kind, bsrc.data_ptr(), bsink.data_ptr(), 8192, 150, st.cuda_stream)
                assert rc == 0, rc
    return queue


with torch.no_grad():
    paths = sorted(glob.glob(os.path.join(ROOT, "tools", "experiments", "head_variants", "*.co")))
    if a.variants:
        paths = [p for p in paths if os.path.basename(p)[:-3] in a.variants.split(",")]
    victims = [(os.path.basename(p)[:-3], launcher(load_variant(p))) for p in paths]
    want = ops.upsample_softargmin(cost, 4 * D, 4 * H, 4 * W)
    refs = {}
    for v, fn in victims:
        r0 = fn().clone()
        torch.cuda.synchronize()
        idle = sum(int((fn().view(torch.int32) != r0.view(torch.int32)).sum()) for _ in range(5))
        refs[v] = r0
        print(f"[{v:12s}] idle GPU: {idle} differing words in 5 launches; max |diff| vs the shipped head {float((r0 - want).abs().max()):.3g} px "
              f"({int((r0.view(torch.int32) != want.view(torch.int32)).sum())} words differ)", flush=True)
    for ld in a.loads.split(","):
        queue = make_load(ld)
        for v, fn in victims:
            torch.cuda.synchronize()
            outs = []
            for it in range(a.iters):
                if queue is not None:
                    queue()
                outs.append(fn())
            torch.cuda.synchronize()
            ref = refs[v].view(torch.int32).flatten()
            bad = [int((o.view(torch.int32).flatten() != ref).sum()) for o in outs]
            print(f"[load={ld:6s}] {v:12s}: {sum(bad):8d} differing words in {sum(1 for b_ in bad if b_):3d} of {a.iters} launches", flush=True)
