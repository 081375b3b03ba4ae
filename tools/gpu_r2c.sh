#!/bin/bash
# round 2: MFMA blur iteration -- blur parity tests, then a short bench
set -u
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -x -q --timeout 120 --timeout-method thread -k "vector_alus or strip_lengths or bench_shape or ragged" > gpurun_out/pytest_mfma.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_mfma.log
[ $rc -ne 0 ] && { grep -E "^(E|FAILED)" gpurun_out/pytest_mfma.log | head -20; exit $rc; }
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end "$@" > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err || { tail -5 gpurun_out/bench_r2c.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bench_r2c.json"))
print("fps %.0f  ms/step %.3f checked %d" % (d["value"], d["ms_per_step"], d["checked_frames"]), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items()})
print("overlapped", d.get("overlapped"))
for o in d.get("other_configs", []): print(o["workload"], round(o["value"]), o["checked_frames"], {k: round(v,4) for k,v in o["kernel_ms_per_step"].items()})
PY
