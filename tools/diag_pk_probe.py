"""r6: narrowest trigger of the r5 co-residency finding -- a matrix of VICTIMS x LOADS, every victim launched repeatedly on
one stream while two other streams loop the load, every output compared bit for bit with the victim's idle-GPU result.

    python tools/diag_pk_probe.py [--iters 30] [--victims ...] [--loads ...]

victims: p<mode>[r<regs>]  single-instruction-form probes of tools/experiments/pk_probe.hip (mode 1..11, regs 0 / 1 / 2 = 8 / 4 / 2 waves per SIMD)
         head_packed        the x4 fused head built WITH packed-fp32 math (openstereo_amd/lib/variants/head_packed.so, through the C ABI)
         head               the shipped head
loads:   none | march (the library's d-marching conv, split in / split out) | brick (the same layer on the brick kernel: OSA_MARCH=0 needs the
         experiments build, so here: the f16 mode of the same layer) | b0..b6 (the synthetic burners of pk_probe.hip)
Prints one line per cell and the identity of the GPU the run was on."""
import argparse
import ctypes as C
import os
import subprocess
import sys

import torch
import torch.nn as nn

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--victims", default="head_packed,head,p1,p2,p3,p4,p5,p6,p7,p8,p9,p10,p11,p4r1,p4r2,p6r1,p6r2")
ap.add_argument("--loads", default="none,march,brick,b0,b1,b2,b3,b4,b5,b6")
ap.add_argument("--probe-iters", type=int, default=96)
a = ap.parse_args()
from openstereo_amd import _lib, engine, ops  # noqa: E402
from openstereo_amd.engine import PackedConv3d  # noqa: E402

_lib.load()
probe = C.CDLL(os.path.join(ROOT, "tools", "experiments", "libpk_probe.so"))
probe.pk_probe_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
probe.burner_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
probe.pk_micro_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
hp_path = os.path.join(ROOT, "openstereo_amd", "lib", "variants", "head_packed.so")
headp = C.CDLL(hp_path) if os.path.exists(hp_path) else None
if headp is not None:
    headp.osa_upsample_softargmin_f32.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]

dev = torch.device("cuda", 0)
try:
    ident = subprocess.run("hostname; rocm-smi --showuniqueid --showserial 2>/dev/null | grep -i -E 'unique|serial' | head -4", shell=True,
                           capture_output=True, text=True).stdout.strip().replace("\n", " | ")
except Exception:
    ident = "?"
print(f"[box] {torch.cuda.get_device_name(0)} | {ident}", flush=True)

g = torch.Generator().manual_seed(1)
B, D, H, W = 3, 48, 136, 240
N = B * 4 * H * 4 * W
cost = (torch.randn(B, D, H, W, generator=g) * 3.0).to(dev)
pin = torch.randn(N, generator=g).to(dev)
micro_in = (torch.randn(D * H * W, generator=g) * 3.0).to(dev)
bsrc = torch.randn(1 << 16, generator=g).to(dev)
bsink = torch.zeros(256, device=dev)
side = [torch.cuda.Stream(), torch.cuda.Stream()]


def cur():
    return torch.cuda.current_stream().cuda_stream


def make_victim(name):
    if name == "head":
        return lambda: ops.upsample_softargmin(cost, 4 * D, 4 * H, 4 * W)
    if name == "head_packed":
        if headp is None:
            return None

        def run():
            out = torch.empty(B, 4 * H, 4 * W, device=dev)
            rc = headp.osa_upsample_softargmin_f32(cost.data_ptr(), out.data_ptr(), B, D, H, W, 4 * D, 4 * H, 4 * W, 0, cur())
            assert rc == 0
            return out
        return run
    if name[0] == "m":                                    # micro probes (third round): four tap loads, two products, ONE packed add per step
        mode = int(name[1:])

        def run_m():
            out = torch.empty(N, device=dev)
            rc = probe.pk_micro_launch(mode, micro_in.data_ptr(), out.data_ptr(), N, D, H * W, cur())
            assert rc == 0, rc
            return out
        return run_m
    assert name[0] == "p"
    mode, _, regs = name[1:].partition("r")
    mode, regs = int(mode), int(regs or 0)

    def run():
        out = torch.empty(N, device=dev)
        rc = probe.pk_probe_launch(mode, regs, pin.data_ptr(), out.data_ptr(), N, a.probe_iters, cur())
        assert rc == 0, rc
        return out
    return run


def make_load(name):
    """returns a function that queues one round of load on both side streams (or None)"""
    if name == "none":
        return None
    if name in ("march", "brick"):
        conv = nn.Conv3d(32, 32, 3, padding=1, bias=False).to(dev)
        if name == "march":
            pc0 = PackedConv3d(conv, None, 1, precision="f16x3")
            run = lambda t: pc0(t, out_split=True)
            xs = []
            for _ in range(2):
                t = ops.to_cl(torch.randn(B, 32, D, H, W, generator=g).to(dev))
                t._osa_meta = engine.input_meta(t)
                xs.append(pc0(t, out_split=True))
            lib = _lib.load()
            n0 = lib.osa_conv3d_march_launches()
            run(xs[0])
            assert lib.osa_conv3d_march_launches() == n0 + 1, "the load must be the d-marching form"
        else:
            pc0 = PackedConv3d(conv, None, 1, precision="f16")
            run = pc0
            xs = [ops.to_cl(torch.randn(B, 32, D, H, W, generator=g).to(dev)) for _ in range(2)]
        torch.cuda.synchronize()

        def queue():
            for st, t in zip(side, xs):
                with torch.cuda.stream(st):
                    for _ in range(3):
                        run(t)
        return queue
    kind = int(name[1:])

    def queue():
        for st in side:
            for _ in range(3):
                rc = probe.burner_launch(kind, bsrc.data_ptr(), bsink.data_ptr(), 8192, 150, st.cuda_stream)
                assert rc == 0, rc
    return queue


with torch.no_grad():
    victims = [(v, make_victim(v)) for v in a.victims.split(",")]
    refs = {}
    for v, fn in victims:
        if fn is None:
            print(f"[{v}] not built -- skipped")
            continue
        r0 = fn().clone()
        torch.cuda.synchronize()
        idle = sum(int((fn().view(torch.int32) != r0.view(torch.int32)).sum()) for _ in range(5))
        refs[v] = r0
        if idle:
            print(f"[{v}] NOT deterministic on an idle GPU: {idle} differing words in 5 launches")
    for ld in a.loads.split(","):
        queue = make_load(ld)
        # how long one round of this load takes (so that cells are comparable)
        t_load = 0.0
        if queue is not None:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side[0])
            queue()
            e1.record(side[0])
            torch.cuda.synchronize()
            t_load = e0.elapsed_time(e1)
        for v, fn in victims:
            if fn is None:
                continue
            torch.cuda.synchronize()
            outs = []
            for it in range(a.iters):
                if queue is not None:
                    queue()
                outs.append(fn())
            torch.cuda.synchronize()
            ref = refs[v].view(torch.int32).flatten()
            bad = [int((o.view(torch.int32).flatten() != ref).sum()) for o in outs]
            extra = ""
            if sum(bad):
                o = outs[[i for i, b_ in enumerate(bad) if b_][0]]
                idx = (o.view(torch.int32).flatten() != ref).nonzero().flatten()
                runs, start = [], None
                il = idx.tolist()
                for j, x in enumerate(il):
                    if start is None:
                        start, prev = x, x
                    elif x == prev + 1:
                        prev = x
                    else:
                        runs.append((start, prev - start + 1)); start, prev = x, x
                runs.append((start, prev - start + 1))
                lens = sorted(set(r[1] for r in runs))
                al = sorted(set(r[0] % 16 for r in runs))
                extra = f"; first launch: {len(runs)} runs, lengths {lens[:6]}, start mod 16 {al[:6]}, max |diff| {float((o.flatten() - refs[v].flatten()).abs().max()):.3g}"
            print(f"[load={ld:6s} {t_load:6.2f} ms/round] {v:12s}: {sum(bad):8d} differing words in {sum(1 for b_ in bad if b_):3d} of {a.iters} launches{extra}", flush=True)
