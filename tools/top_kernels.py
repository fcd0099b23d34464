"""Top kernels of a rocprofv3 --kernel-trace --stats run:  python tools/top_kernels.py <..._kernel_stats.csv> [steps] [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step: {tot / steps / 1e6:.3f} ms over {steps} steps")
for r in rows[:n]:
    print(f"{float(r['TotalDurationNs']) / tot * 100:5.1f}% {int(r['Calls']) / steps:7.1f}/step avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:120]}")
