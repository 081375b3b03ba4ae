#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate PMC passes of bench.py.
# usage: tools/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/{trace,fetch,write,sq}
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-overlap"   # the plain one-stream loop that `value` and the roofline are measured on
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps 20 --warmup 3 > "$OUT/trace.json" 2> "$OUT/trace.err" || { echo "trace pass failed"; tail -5 "$OUT/trace.err"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BENCH --steps 2 --warmup 1 --no-profile-events > "$OUT/fetch.json" 2> "$OUT/fetch.err" || { echo "fetch pass failed"; tail -5 "$OUT/fetch.err"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BENCH --steps 2 --warmup 1 --no-profile-events > "$OUT/write.json" 2> "$OUT/write.err" || { echo "write pass failed"; tail -5 "$OUT/write.err"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/sq" -- $BENCH --steps 2 --warmup 1 --no-profile-events > "$OUT/sq.json" 2> "$OUT/sq.err" || { echo "sq pass failed"; tail -5 "$OUT/sq.err"; exit 1; }
find "$OUT" -name '*.csv' | head -20
