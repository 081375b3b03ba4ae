#!/bin/bash
# GPU box: bench.py once per value of an environment variable (tuning knobs of the library).
# usage: tools/sweep_env.sh VAR v1 v2 ... [-- bench args]
VAR=$1; shift
VALS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for v in "${VALS[@]}"; do
  env $VAR=$v timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/sweep_$v.json 2> gpurun_out/sweep_$v.err || { tail -3 gpurun_out/sweep_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/sweep_$v.json"))
print("$VAR=%-8s fps %.0f" % ("$v", d["value"]), {k: round(x,4) for k,x in d["kernel_ms_per_step"].items()})
PY
done
