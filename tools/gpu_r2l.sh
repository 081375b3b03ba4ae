#!/bin/bash
set -u
timeout -k 10 400 python -u -m pytest tests -x -q --timeout 180 --timeout-method thread -m gpu -k "batch or ingest or widths or bench_line" > gpurun_out/pytest_batch.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_batch.log
[ $rc -ne 0 ] && { grep -E "^(E|FAILED)" gpurun_out/pytest_batch.log | head; exit $rc; }
for b in 256 64; do
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --batch $b > gpurun_out/bench_r2l_$b.json 2> gpurun_out/bench_r2l.err || { tail -5 gpurun_out/bench_r2l.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bench_r2l_$b.json"))
print("batch $b fps %.0f" % d["value"], {k: round(v,4) for k,v in d["kernel_ms_per_step"].items()}, d["roofline"]["traffic"], d.get("roofline_binding"))
e=d["end_to_end"]; print(round(e["sift_hip_batch"]["value"]), e["sift_hip_batch"]["ms_per_call"], round(e["sift_hip_batch_top"]["value"]), e["sift_hip_batch_top"]["ms_per_call"])
PY
done
