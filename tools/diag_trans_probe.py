"""r5 probe (tools/experiments/trans_probe.hip): which instruction class of a VALU kernel returns wrong results while f16 MFMA convolutions run
on other streams?  python tools/diag_trans_probe.py [--load f16x3|f16|f32|gemm16|none]"""
import argparse
import ctypes as C
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--load", default="f16x3")
ap.add_argument("--iters", type=int, default=40)
a = ap.parse_args()
from openstereo_amd import _lib, engine, ops  # noqa: E402
from openstereo_amd.engine import PackedConv3d  # noqa: E402
_lib.load()
lib = C.CDLL(os.path.join(ROOT, "tools", "experiments", "libtrans_probe.so"))
lib.probe_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
PL, NP = 136 * 240, 48
inp = (torch.randn(NP * PL, generator=g) * 3.0).to(dev)
N = 3 * 544 * 960
names = {1: "v_exp_f32 (registers)", 2: "global loads (head pattern)", 3: "fma only", 4: "v_rcp_f32", 5: "loads + v_exp_f32 (head-like)", 6: "polynomial exp2, no trans op"}
with torch.no_grad():
    loads, run_load = [], None
    if a.load in ("f16x3", "f16", "f32"):
        conv = nn.Conv3d(32, 32, 3, padding=1, bias=False).to(dev)
        pc = PackedConv3d(conv, None, 1, precision=a.load)
        xs = [ops.to_cl(torch.randn(3, 32, 48, 136, 240, generator=g).to(dev)) for _ in range(2)]
        for t in xs:
            if a.load == "f16x3":
                t._osa_meta = engine.input_meta(t)
        loads = [(torch.cuda.Stream(), t) for t in xs]
        run_load = lambda t: pc(t)
    elif a.load == "gemm16":
        A = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
        loads = [(torch.cuda.Stream(), A), (torch.cuda.Stream(), A.clone())]
        run_load = lambda t: t @ t
    for mode in (1, 2, 3, 4, 5, 6):
        def launch():
            out = torch.empty(N, device=dev)
            rc = lib.probe_launch(inp.data_ptr(), out.data_ptr(), N, PL, NP, mode, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            return out
        torch.cuda.synchronize()
        ref = launch().clone()
        torch.cuda.synchronize()
        outs = []
        for it in range(a.iters):
            for st, t in loads:
                with torch.cuda.stream(st):
                    for _ in range(3):
                        run_load(t)
            outs.append(launch())
        torch.cuda.synchronize()
        bad = [int((o.view(torch.int32) != ref.view(torch.int32)).sum()) for o in outs]
        first = ""
        if sum(bad):
            o = outs[[i for i, b in enumerate(bad) if b][0]]
            idx = (o.view(torch.int32) != ref.view(torch.int32)).nonzero().flatten()
            runs = idx[:40].tolist()
            first = f"; first differing indices {runs[:12]} (index mod 64: {[i % 64 for i in runs[:12]]})"
        print(f"[load={a.load}] mode {mode} {names[mode]:32s}: {sum(bad)} differing elements in {sum(1 for b in bad if b)} of {a.iters} launches{first}")
