#!/bin/bash
# GPU box: like ab.sh, but with the other configurations of bench.py (sparse and small frames react differently).
# usage: tools/ab_configs.sh <lib1.so> <lib2.so> ...
for r in 1 2; do for src in "$@"; do
v=$(basename "$src" .so)
cp "$src" sift_amd/lib/libsift_hip.so || exit 1
timeout -k 10 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-end-to-end > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -3 gpurun_out/ab_$v.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
print("%-10s" % "$v", round(d["value"]), round(d["kernel_ms_per_step"]["k_describe"],4), [(round(o["value"]), round((o.get("kernel_ms_per_step") or {}).get("k_describe",0),4)) for o in d["other_configs"]])
PY
done; done
