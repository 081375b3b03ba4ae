#!/bin/bash
# GPU box: A/B two builds of the library in one call (same box, same clocks).
# usage: tools/ab.sh <libA.so> <libB.so> [bench args]   -> alternates A,B,A,B and prints kernel ms
A=$1; B=$2; shift 2
for r in 1 2; do for v in A B; do
  eval "src=\$$v"; cp "$src" sift_amd/lib/libsift_hip.so
  timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -3 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
print("$v fps %.0f" % d["value"], {k: round(x,4) for k,x in d["kernel_ms_per_step"].items()})
PY
done; done
