#!/bin/bash
# GPU box: how busy the units are in each kernel (separate counter passes of a short bench run).
# usage: tools/pmc_units.sh <tag>   -> gpurun_out/units_<tag>/{ta,sq,lds}
set -u
TAG=${1:-u}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/units_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-overlap --steps 2 --warmup 1 --no-profile-events --batch 64"
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d "$OUT/ta" -- $BENCH > "$OUT/ta.json" 2> "$OUT/ta.err" || { echo "ta pass failed"; tail -5 "$OUT/ta.err"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/sq" -- $BENCH > "$OUT/sq.json" 2> "$OUT/sq.err" || { echo "sq pass failed"; tail -5 "$OUT/sq.err"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/lds" -- $BENCH > "$OUT/lds.json" 2> "$OUT/lds.err" || { echo "lds pass failed"; tail -5 "$OUT/lds.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("ta", "sq", "lds"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (out, p)):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "sift::k_" not in n:
                continue
            k = n.split("sift::")[1].split("<")[0].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        print(p, k, {c: round(sum(v) / len(v)) for c, v in agg[k].items()})
PY
