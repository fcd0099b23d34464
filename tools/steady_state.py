"""Steady-state view of a rocprofv3 --kernel-trace run of a periodic workload (training): kernels of the LAST `window_ms` only, so warm-up
(MIOpen find-mode candidates, weight packing) does not pollute the table.
    python tools/steady_state.py <..._kernel_trace.csv> <window_ms> <steps_in_window> [n]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
win, steps = float(sys.argv[2]) * 1e6, int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
t_end = max(int(r["End_Timestamp"]) for r in rows)
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - win]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
print(f"window {win / 1e6:.0f} ms = {steps} steps: {len(sel) / steps:.0f} kernel launches and {busy / steps / 1e6:.1f} ms of kernel time per step "
      f"({busy / win * 100:.0f} % of the window)")
agg = defaultdict(lambda: [0, 0])
for r in sel:
    a = agg[r["Kernel_Name"]]
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print(f"{t / busy * 100:5.1f}% {c / steps:8.1f}/step avg {t / c / 1e3:9.1f} us  {name[:110]}")
