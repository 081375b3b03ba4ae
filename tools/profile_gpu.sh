#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate PMC passes of bench.py's plain one-stream loop.
# usage: tools/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/{trace,fetch,write,sq,ta,lds}
# Summarise afterwards (here) with: python tools/pmc_traffic.py gpurun_out/prof_<tag> <tag> --workload 1920x1080x256
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
# the kernel sources this build was made from (bench.py quotes the counters only for the same sources)
python3 - "$ROOT" > "$OUT/kernel_sources.sha256" <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
import bench
print(bench.kernel_sources_sha256())
PY
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-overlap --no-other-configs --no-end-to-end"   # the loop that `value` and the roofline are measured on
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps 20 --warmup 3 > "$OUT/trace.json" 2> "$OUT/trace.err" || { echo "trace pass failed"; tail -5 "$OUT/trace.err"; exit 1; }
pass() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- $BENCH --steps 2 --warmup 1 --no-profile-events > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "$name pass failed"; tail -5 "$OUT/$name.err"; exit 1; }
}
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
pass write WRITE_SIZE || exit 1
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE || exit 1
pass ta TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE || exit 1
pass lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE || exit 1
find "$OUT" -name '*.csv' | head -20
