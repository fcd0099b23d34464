"""rocprofv3 --pmc counter_collection.csv -> one row per (kernel, grid, counter): dispatches and the average counter value (the compact form
kept under profiles/roundN/; tools/parse_pmc3.py reads the raw files).

    python tools/aggregate_pmc.py gpurun_out/prof_r4b/pmc_FETCH_SIZE > profiles/round4/r4_final_pmc_FETCH_SIZE_B9.csv
"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0), r["Counter_Name"])].append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Grid_Size", "Counter_Name", "Dispatches", "Average_Counter_Value"])
for (k, g, c), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([k, g, c, len(v), sum(v) / len(v)])
