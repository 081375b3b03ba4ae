#!/bin/bash
# GPU box: one rocprofv3 --pmc pass of bench.py; prints the per-kernel mean of every counter.
# usage: tools/pmc.sh <tag> <counter> [<counter> ...]   (one pass; respect the per-block slot limits)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-profile-events ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/err.txt" || { echo "pmc pass failed"; tail -5 "$OUT/err.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        m = re.search(r"sift::(k_[a-z_0-9]+)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(acc[k].items())})
PY
