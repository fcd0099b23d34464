"""Idle gaps between consecutive kernels of a periodic workload (rocprofv3 --kernel-trace CSV): where the wall time of a replayed training
step goes when its kernel time is half of it.    python tools/gap_analysis.py <kernel_trace.csv> <window_ms> [n]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6
n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
t_end = max(int(r["End_Timestamp"]) for r in rows)
sel = sorted((r for r in rows if int(r["Start_Timestamp"]) >= t_end - win), key=lambda r: int(r["Start_Timestamp"]))
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
gaps, after, before, hist = [], defaultdict(lambda: [0, 0]), defaultdict(lambda: [0, 0]), defaultdict(int)
cur_end = int(sel[0]["End_Timestamp"])
for prev, r in zip(sel, sel[1:]):
    g = int(r["Start_Timestamp"]) - cur_end
    cur_end = max(cur_end, int(r["End_Timestamp"]))
    if g <= 0:
        continue
    gaps.append(g)
    a, b = after[prev["Kernel_Name"][:90]], before[r["Kernel_Name"][:90]]
    a[0] += 1; a[1] += g; b[0] += 1; b[1] += g
    hist[min(int(g / 2000), 20)] += 1
tot = sum(gaps)
print(f"window {span / 1e6:.1f} ms: {len(sel)} kernels, busy {busy / 1e6:.1f} ms, idle gaps {tot / 1e6:.1f} ms in {len(gaps)} gaps (median {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us)")
print("gap histogram (2 us bins, last = 40 us and more):", [hist[i] for i in range(21)])
print("idle time by the kernel that FOLLOWS the gap:")
for k, (c, t) in sorted(before.items(), key=lambda kv: -kv[1][1])[:n]:
    print(f"  {t / tot * 100:5.1f}%  {c:6d} gaps  avg {t / c / 1e3:7.1f} us  {k}")
print("idle time by the kernel that PRECEDES the gap:")
for k, (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:n]:
    print(f"  {t / tot * 100:5.1f}%  {c:6d} gaps  avg {t / c / 1e3:7.1f} us  {k}")
