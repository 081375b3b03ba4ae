#!/bin/bash
# GPU box: compare builds of the library in one call (same box, same clocks).
# usage: tools/ab.sh <lib1.so> <lib2.so> [<lib3.so> ...] [-- bench args]
#   runs bench.py with each build in turn, twice round, and prints the kernel ms per step.
# The builds must lie in a directory that travels with the snapshot (ab/, git-ignored) --
# gpurun_out/ does not.  The in-tree library is overwritten: restore it afterwards.
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for r in 1 2; do for src in "${LIBS[@]}"; do
  v=$(basename "$src" .so)
  cp "$src" sift_amd/lib/libsift_hip.so || { echo "ab.sh: cannot copy $src"; exit 1; }
  timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-other-configs --no-end-to-end "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -3 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
print("%-14s fps %.0f" % ("$v", d["value"]), {k: round(x,4) for k,x in d["kernel_ms_per_step"].items()})
PY
done; done
