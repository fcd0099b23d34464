"""Per-kernel HBM traffic from the rocprofv3 PMC passes of tools/profile_round2.sh -> JSON for profiles/traffic.json.
bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB; FETCH_SIZE reads half of a wide coalesced stream on gfx950 -- MI355X_MICROARCH.md
'HBM'), averaged over the dispatches of that kernel in the run (all at the bench batch: --timed-only)."""
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 8      # pairs per step of the profiled bench run (bench.py default)
KERNELS = {   # key in traffic.json -> substring of the rocprof kernel name (+ optional grid filter)
    "conv3d_32_32_V0_f16x3": "conv_mfma_kernel<1, 1, 1, 2, 1, 4, 1, 8, 8, 0, 1, 0, 1>",
    "volume": "build_volume_quads_kernel<2, 8>",
    "head": "upsample4_softargmin_kernel",
    "classifier": "conv_small_co_tiled_kernel<1, true>",
}


def per_kernel(counter):
    acc = {}
    for f in glob.glob(os.path.join(out_dir, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True):
        rows = sorted((r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == counter), key=lambda r: int(r["Dispatch_Id"]))
        for row in rows:
            name, grid = row["Kernel_Name"], int(row.get("Grid_Size", 0) or 0)
            acc.setdefault((name, grid), []).append(float(row["Counter_Value"]))
    return acc


fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
res = {"_note": f"HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc passes over `bench.py --timed-only --no-graph` "
                f"(f16x3, {BATCH} pairs per step); per kernel instance, the LARGEST grid of that kernel (the 48x136x240 / full-size launches), averaged over "
                "its dispatches.  The x2 on FETCH_SIZE is the guide's gfx950 correction, calibrated on wide coalesced streams (it reproduces the volume "
                "builder's algorithmic bytes within 0.1 %); the conv / classifier staging reads 64-byte segments at a 128-byte stride, for which the "
                "correction is uncalibrated -- _detail carries the undoubled figure too.  Source: profiles/round2/pmc_*.csv (tools/profile_round2.sh, "
                "tools/parse_pmc.py)."}
detail = {}
for key, sub in KERNELS.items():
    fk = {k: v for k, v in fetch.items() if sub in k[0]}
    wk = {k: v for k, v in write.items() if sub in k[0]}
    if not fk or not wk:
        continue
    g = max(k[1] for k in fk)                       # the full-resolution launches
    f = [v for k, v in fk.items() if k[1] == g][0]
    w = [v for k, v in wk.items() if k[1] == g][0]
    if key.startswith("conv3d_32_32"):
        # this instance runs 4 times per step in dispatch order: dres0.0 (64 -> 32), dres0.2, dres1.0 (32 -> 32, no residual), dres1.2
        # (32 -> 32 + residual); the roofline entry is the plain 32 -> 32 launch
        f = [v for i, v in enumerate(f) if i % 4 in (1, 2)]
        w = [v for i, v in enumerate(w) if i % 4 in (1, 2)]
    fb, wb = sum(f) / len(f) * 1024.0, sum(w) / len(w) * 1024.0
    res[key + f"_B{BATCH}"] = int(2 * fb + wb)
    detail[key] = {"grid": g, "dispatches": len(f), "fetch_size_kib_avg": round(sum(f) / len(f), 1), "write_size_kib_avg": round(sum(w) / len(w), 1),
                   "bytes_if_fetch_size_is_not_doubled": int(fb + wb)}
res["_detail"] = detail
print(json.dumps(res, indent=1))
