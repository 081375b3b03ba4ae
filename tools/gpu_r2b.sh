#!/bin/bash
# round 2: MFMA blur -- operand layout check, parity tests, bench (new bench.py with all its passes)
set -u
timeout -k 10 60 tools/ubench/mfma_i8_layout || exit 1
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -x -v --timeout 120 --timeout-method thread -k "vector_alus or strip_lengths or bench_shape" > gpurun_out/pytest_mfma.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_mfma.log
[ $rc -ne 0 ] && { grep -E "^(E|FAILED)" gpurun_out/pytest_mfma.log | head -20; exit $rc; }
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_r2b.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_r2b.log
[ $rc -ne 0 ] && { grep -E "^(E|FAILED)" gpurun_out/pytest_r2b.log | head -20; exit $rc; }
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err || { tail -5 gpurun_out/bench_r2b.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bench_r2b.json"))
print("fps %.0f  ms/step %.3f checked %d" % (d["value"], d["ms_per_step"], d["checked_frames"]), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items()})
print("overlapped", d.get("overlapped"))
for o in d.get("other_configs", []): print(o["workload"], round(o["value"]), o["checked_frames"], {k: round(v,4) for k,v in o["kernel_ms_per_step"].items()})
print(d.get("end_to_end"))
print(d.get("cpu_baseline"))
PY
