# usage: pmc_layer.sh TAG -- <command>     kernel-trace + FETCH / WRITE / SQ passes of one command; prints per-kernel averages
TAG=$1; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- "$@" > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- "$@" > $OUT/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- "$@" > $OUT/sq.log 2>&1
python - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
def rows(sub):
    acc = {}
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc.setdefault((r["Kernel_Name"][:90], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return acc
stats = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"][:90]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
fe, wr, sq = rows("fetch"), rows("write"), rows("sq")
for k, (calls, ms) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:6]:
    f = fe.get((k, "FETCH_SIZE"), [0]); w = wr.get((k, "WRITE_SIZE"), [0])
    fb, wb = sum(f) / len(f) * 1024 * 2, sum(w) / len(w) * 1024
    g = lambda c: (lambda v: sum(v) / len(v) if v else 0)(sq.get((k, c), []))
    busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / max(g("SQ_BUSY_CYCLES"), 1) / 4 if g("SQ_BUSY_CYCLES") else 0
    wc = max(g("SQ_WAVE_CYCLES"), 1)
    print(f"{k[:70]:70s} calls {calls:4d} avg {ms:7.3f} ms | HBM read {fb/1e6:8.1f} MB (2 x FETCH_SIZE) write {wb/1e6:8.1f} MB total {(fb+wb)/1e6:8.1f} MB = {(fb+wb)/ms/1e9:6.2f} TB/s | "
          f"MFMA busy {g('SQ_VALU_MFMA_BUSY_CYCLES'):.3g} of SQ_BUSY {g('SQ_BUSY_CYCLES'):.3g} | wait_any {g('SQ_WAIT_ANY')/wc:.2f} wait_inst {g('SQ_WAIT_INST_ANY')/wc:.2f} active {g('SQ_ACTIVE_INST_ANY')/wc:.2f} of wave cycles")
PY
