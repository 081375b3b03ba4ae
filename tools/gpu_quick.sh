#!/bin/bash
# GPU box: parity tests then a bench line.  usage: tools/gpu_quick.sh <tag> [bench args]
TAG=${1:-q}; shift
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_$TAG.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_$TAG.log
[ $rc -ne 0 ] && { grep -E "^(E|FAILED)" gpurun_out/pytest_$TAG.log | head -20; exit $rc; }
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$TAG.json"))
print("fps %.0f  ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items()})
PY
