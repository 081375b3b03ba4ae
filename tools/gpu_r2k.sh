#!/bin/bash
set -u
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r2k.json 2> gpurun_out/bench_r2k.err || { tail -5 gpurun_out/bench_r2k.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bench_r2k.json"))
print("fps %.0f  ms/step %.3f checked %d" % (d["value"], d["ms_per_step"], d["checked_frames"]), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items()})
PY
bash tools/profile_gpu.sh ${1:-r02b}
