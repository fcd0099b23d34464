#!/bin/bash
# kernel table + idle gaps of a training workload as it is TIMED (hipGraph replay): bash tools/prof_train_graph.sh <workload> <tag> <window_ms> <steps_in_window> [extra bench.py flags, e.g. --amp]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$1; TAG=$2; WIN=$3; NS=$4; EXTRA="${@:5}"
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $R/bench.py --workload $W --timed-only --steps 12 --warmup 4 $EXTRA > $OUT/stdout.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/steady_state.py $F $WIN $NS 45 > $OUT/steady_state.txt 2>&1
python - "$F" > $OUT/gaps.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-3000:]
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
gaps_s = sorted(gaps)
print("last 3000 kernels: median gap %.1f us, mean %.1f us, p90 %.1f us, busy %.1f ms, span %.1f ms" % (
    gaps_s[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3, gaps_s[int(len(gaps) * 0.9)] / 1e3,
    sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6, (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6))
big = sorted(zip(gaps, [(a["Kernel_Name"][:60], b["Kernel_Name"][:60]) for a, b in zip(rows, rows[1:])]), reverse=True)[:12]
for g, (a, b) in big:
    print("%8.1f us  after %s -> before %s" % (g / 1e3, a, b))
PY
rm -rf $OUT/trace
tail -2 $OUT/stdout.log
