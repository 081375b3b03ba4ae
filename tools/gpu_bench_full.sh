#!/bin/bash
# GPU box: the default bench.py run (what the driver runs), timed.  usage: tools/gpu_bench_full.sh <tag>
TAG=${1:-full}
t0=$(date +%s)
timeout -k 10 400 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
echo "bench.py took $(( $(date +%s) - t0 )) s"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$TAG.json"))
print("fps %.0f  ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items()})
print("roofline", {k: d["roofline"][k] for k in ("kernel","frac","traffic")}, "cpu", d["cpu_baseline"]["value"])
for o in d.get("other_configs", []):
    print(o["workload"][:60], "fps %.0f" % o["value"], o.get("kernel_ms_per_step"), o.get("cpu_baseline", {}).get("value"), o["checked_frames"])
e=d.get("end_to_end") or {}
for k,v in e.items():
    if isinstance(v, dict) and "value" in v: print(k, "fps %.0f" % v["value"], v["ms_per_call"])
    if k == "sparse": print("sparse:", {kk: round(vv["value"]) for kk, vv in v.items() if isinstance(vv, dict)})
PY
