"""rocprofv3 PMC passes of tools/profile_round3.sh -> traffic.json (HBM bytes per launch + `_calibration`) on stdout and a per-kernel SQ
table (MFMA busy, waits) in <out_dir>/sq_summary.txt."""
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 8
KERNELS = {
    # r4: the V0 3x3x3 32 -> 32 layers run the d-marching kernel; its <split out, split in> instance runs dres0.2, dres1.0, dres1.2
    # (dres0.0 reads the fp32 volume: <.., 1, 0>; classif3.0 writes fp32: <.., 0, 1>).  r3 and before: the brick kernel instance below.
    "conv3d_32_32_V0_f16x3": os.environ.get("OSA_PMC_DOMINANT", "conv_march_kernel<4, 16, 1, 1>"),
    "conv3d_32_32_V0_f16x3_brick": "conv_mfma_kernel<1, 1, 1, 2, 1, 4, 1, 8, 8, 0, 1, 0, 1, 0>",
    # (r4, later: every conv_mfma_kernel name carries a 14th template argument, BL -- 1 = weight fragments through the LDS ring)
    "volume": "build_volume_walk_kernel<2, 8, ",
    "head": "upsample4_softargmin_kernel",
    "classifier": "classifier_march_kernel",
    "deconv_64_32_redir": "conv_mfma_kernel<1, 8, 1, 1, 1, 4, 1, 4, 8, 1, 1, 0, 1, 0>",
    "deconv_128_64_redir": "conv_mfma_kernel<1, 8, 1, 1, 1, 4, 1, 4, 8, 2, 1, 0, 1, 0>",
    "conv_s2_32_64": os.environ.get("OSA_PMC_S2", "conv_march_s2_kernel<8>"),          # r6: the stride-2 d-marching form (before: conv_mfma_kernel<1, 1, 3, 1, 1, 2, 2, 4, 8, 0, 1, 0, 1, 0>)
    "conv_s2_64_128": "conv_mfma_kernel<1, 1, 3, 1, 1, 2, 2, 4, 8, 0, 1, 0, 1, 0>",
    "backbone_64ch_quarter": "conv_mfma_kernel<1, 1, 1, 1, 2, 4, 1, 8, 16, 0, 1, 0, 1, 1>",
    "backbone_128ch_quarter": "conv_mfma_kernel<1, 1, 3, 2, 2, 2, 2, 8, 16, 0, 1, 0, 1, 0>",
}


def rows_of(sub, counter=None):
    acc = {}
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if counter and row.get("Counter_Name") != counter:
                continue
            acc.setdefault((row["Kernel_Name"], int(row.get("Grid_Size", 0) or 0), row["Counter_Name"]), []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in acc.items()}


fetch, write = rows_of("pmc_FETCH_SIZE", "FETCH_SIZE"), rows_of("pmc_WRITE_SIZE", "WRITE_SIZE")
res = {"_note": f"HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc passes over `bench.py --timed-only --no-graph` "
                f"(f16x3, {BATCH} pairs per step), per kernel instance the LARGEST grid, averaged over its dispatches; the x2 on FETCH_SIZE is the "
                "MI355X_MICROARCH.md gfx950 correction -- see `_calibration` for what it does on the engine's own access patterns "
                "(tools/profile_round3.sh, tools/parse_pmc3.py, tools/calibrate_traffic.py)."}
detail = {}
for key, sub in KERNELS.items():
    fk = {k: v for k, v in fetch.items() if sub in k[0]}
    wk = {k: v for k, v in write.items() if sub in k[0]}
    if not fk or not wk:
        continue
    g = max(k[1] for k in fk)
    f = [v for k, v in fk.items() if k[1] == g][0]
    w = [v for k, v in wk.items() if k[1] == g][0]
    if key == "conv3d_32_32_V0_f16x3" and "conv_march_kernel<4, 16, 1, 1>" in sub and len(f) % 4 == 0 and os.environ.get("OSA_PMC_VOLUME_SPLIT", "1") == "1":
        # late r4: the volume is a split tensor, so dres0.0 (64 -> 32) runs this instance too -- dispatch order per step dres0.0, dres0.2, dres1.0,
        # dres1.2: price the three 32 -> 32 launches (OSA_PMC_VOLUME_SPLIT=0: traces taken with OSA_VOL_SPLIT=0 / before that change)
        f = [v for i, v in enumerate(f) if i % 4 != 0]
        w = [v for i, v in enumerate(w) if i % 4 != 0]
    if key.startswith("conv3d_32_32") and "conv_mfma_kernel" in sub:   # brick instance, dispatch order per step: dres0.0 (64 -> 32), dres0.2, dres1.0, dres1.2: price the plain 32 -> 32 launches
        f = [v for i, v in enumerate(f) if i % 4 in (1, 2)]
        w = [v for i, v in enumerate(w) if i % 4 in (1, 2)]
    fb, wb = sum(f) / len(f) * 1024.0, sum(w) / len(w) * 1024.0
    res[key + f"_B{BATCH}"] = int(2 * fb + wb)
    detail[key] = {"grid": g, "dispatches": len(f), "fetch_size_kib_avg": round(sum(f) / len(f), 1), "write_size_kib_avg": round(sum(w) / len(w), 1),
                   "bytes_if_fetch_size_is_not_doubled": int(fb + wb)}
res["_detail"] = detail

# ---- calibration: dwconv2d identity over [1, 8192, 8192, 32]: C = 32 (full lines) then C = 16 (64-B segments at a 128-B stride)
cf, cw = rows_of("cal_FETCH_SIZE", "FETCH_SIZE"), rows_of("cal_WRITE_SIZE", "WRITE_SIZE")
cal = {}
npx = 8192 * 8192
for C in (32, 16):
    grid = npx * (C // 4)                              # one thread per (pixel, channel quad); rocprof reports the grid in threads (rounded up to 256)
    fk = [v for k, v in cf.items() if "dwconv2d_nhwc_kernel" in k[0] and abs(k[1] - grid) <= 256]
    wk = [v for k, v in cw.items() if "dwconv2d_nhwc_kernel" in k[0] and abs(k[1] - grid) <= 256]
    if fk and wk:
        f, w = sum(fk[0]) / len(fk[0]) * 1024.0, sum(wk[0]) / len(wk[0]) * 1024.0
        useful = npx * C * 4
        cal[f"read_{C}_of_32_channels"] = {"useful_read_bytes": useful, "FETCH_SIZE_bytes": int(f), "FETCH_SIZE_over_useful": round(f / useful, 4),
                                           "lines_touched_bytes": npx * 128, "FETCH_SIZE_over_lines_touched": round(f / (npx * 128), 4)}
        cal[f"write_{C}_channels_contiguous"] = {"useful_write_bytes": useful, "WRITE_SIZE_bytes": int(w), "WRITE_SIZE_over_useful": round(w / useful, 4)}
cal["_reading"] = ("FETCH_SIZE_over_useful = 0.5 on the full-line stream reproduces the guide's 'x2' rule.  For the half-line pattern: a ratio of 0.5 against "
                   "USEFUL bytes means the x2 rule also holds there (64-byte requests counted as such, traffic = useful bytes); a ratio of 0.5 against LINES "
                   "TOUCHED (i.e. 1.0 against useful) means whole 128-byte lines cross the fabric and the doubled figure is the real traffic.")
res["_calibration"] = cal
print(json.dumps(res, indent=1))

# ---- SQ counters per kernel instance (largest grid)
sq = rows_of("pmc_SQ")
names = sorted({k[0] for k in sq})
with open(os.path.join(out_dir, "sq_summary.txt"), "w") as fh:
    fh.write(f"# rocprofv3 --pmc SQ_* pass over bench.py --timed-only --no-graph, f16x3, {BATCH} pairs per step: per kernel instance (largest grid), average per launch.\n"
             "# SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs (= #MFMA x 32 cycles for v_mfma_f32_32x32x16_f16: checked against the launch's MFMA count);\n"
             "# GRBM_GUI_ACTIVE is summed over the 8 XCDs, so launch cycles = GRBM_GUI_ACTIVE / 8 and\n"
             "# MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs) = fraction of SIMD-cycles with the matrix pipe busy.\n"
             "# (the conv3d_32_32 instance averages its 4 launches per step: dres0.0 with 64 input channels and three 32 -> 32 launches)\n")
    for key, sub in KERNELS.items():
        ks = [k for k in sq if sub in k[0]]
        if not ks:
            continue
        g = max(k[1] for k in ks)
        vals = {k[2]: sum(sq[k]) / len(sq[k]) for k in ks if k[1] == g}
        gui = vals.get("GRBM_GUI_ACTIVE", 0.0)
        busy = vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        share = busy / (gui / 8.0 * 256 * 4) if gui else float("nan")
        wave = vals.get("SQ_WAVE_CYCLES", 0.0)
        fh.write(f"{key:26s} grid {g:9d}  GRBM_GUI_ACTIVE {gui:12.0f}  MFMA_BUSY {busy:14.0f} ({share * 100:5.1f} % of SIMD-cycles)  "
                 f"WAIT_ANY/WAVE {vals.get('SQ_WAIT_ANY', 0) / wave if wave else float('nan'):.2f}  WAIT_INST_ANY/WAVE {vals.get('SQ_WAIT_INST_ANY', 0) / wave if wave else float('nan'):.2f}  "
                 f"ACTIVE_INST_ANY/WAVE {vals.get('SQ_ACTIVE_INST_ANY', 0) / wave if wave else float('nan'):.2f}\n")
