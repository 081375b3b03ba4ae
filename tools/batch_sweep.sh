#!/bin/bash
# GPU box: the device-resident bench at several batch sizes (frames per launch): per-frame kernel times show
# what the batch size costs (tails of short launches) and gains (the pyr plane of a small batch fits the Infinity Cache).
for b in "$@"; do
  timeout -k 10 120 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-end-to-end > gpurun_out/bs_$b.json 2> gpurun_out/bs_$b.err || { tail -3 gpurun_out/bs_$b.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/bs_$b.json"))
b=$b
print("batch %3d: fps %.0f  per frame (us): " % (b, d["value"]), {k: round(1e3*v/b,3) for k,v in d["kernel_ms_per_step"].items()})
PY
done
