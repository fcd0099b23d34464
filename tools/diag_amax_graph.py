"""Diagnostic (GPU): osa_amax_f32 inside a replayed hipGraph."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openstereo_amd import ranges
from openstereo_amd.ranges import input_meta, amax_of
x = torch.randn(3, 32, 17, 33, device="cuda")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        y = x * 2.0
        r = amax_of(input_meta(y)).clone()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = x * 2.0                          # produced inside the graph
    r1 = amax_of(input_meta(y)).clone()
    r2 = amax_of(input_meta(x)).clone()  # static input
    outs = []
    for i in range(300):                 # more blocks than one arena holds
        z = y + float(i)
        outs.append(amax_of(input_meta(z)).clone())
for k in range(4):
    x.mul_(3.0)
    g.replay()
    torch.cuda.synchronize()
    want1, want2 = float((x * 2).abs().max()), float(x.abs().max())
    bad = [i for i in range(300) if float(outs[i]) != float((x * 2 + float(i)).abs().max())]
    print(f"replay {k}: amax(y) {float(r1)!r} want {want1!r}; amax(x) {float(r2)!r} want {want2!r}; mismatching of 300 later blocks: {bad[:10]}")
