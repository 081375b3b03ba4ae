#!/bin/bash
# GPU box: WRITE_SIZE / FETCH_SIZE of the kernels for the in-tree build (two short PMC passes)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-wr}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-overlap --no-other-configs --no-end-to-end --steps 2 --warmup 1 --no-profile-events --no-check"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BENCH > "$OUT/write.json" 2> "$OUT/write.err" || { echo "write pass failed"; tail -5 "$OUT/write.err"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BENCH > "$OUT/fetch.json" 2> "$OUT/fetch.err" || { echo "fetch pass failed"; tail -5 "$OUT/fetch.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("write", "fetch"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for p in glob.glob(out + "/" + name + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void sift::", "").replace("sift::", "")
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in sorted(acc.items()):
        print(name, k, "%.1f MB per launch" % (v / n * 1024 / 1e6))
PY
