#!/bin/bash
# GPU box: kernel trace of a short bench.py run; prints the per-kernel statistics.
# usage: tools/trace_gpu.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" > "$OUT/bench.json" 2> "$OUT/err.txt" || { echo "trace failed"; tail -5 "$OUT/err.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(p)):
        print("%-60s calls %5s avg_us %9.1f pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
