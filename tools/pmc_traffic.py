#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory into profiles/.

  python tools/pmc_traffic.py gpurun_out/prof_<tag> <tag> [--workload 1920x1080x32]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, our kernels),
profiles/<tag>_pmc.json (per-kernel mean of every counter collected in the separate --pmc
passes) and updates profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.

HBM bytes per launch = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB: on gfx950 FETCH_SIZE reports half
the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"sift::(k_[a-z_0-9]+)", name)
    return m.group(1) if m else None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "1920x1080x32"
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        with open(stats[0]) as f, open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"), "w") as g:
            g.write(f.read())
    # launch geometry / registers from the trace
    res = {}
    for p in glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")):
        with open(p) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if k and k not in res:
                    res[k] = {x: int(r[x]) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                                      "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y",
                                                      "Grid_Size_Z")}
    acc = defaultdict(lambda: defaultdict(list))
    for p in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
        with open(p) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    pmc = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
    for k in pmc:
        pmc[k]["_launches"] = max(len(v) for v in acc[k].values())
        if k in res:
            pmc[k]["_resources"] = res[k]
    with open(os.path.join(ROOT, "profiles", tag + "_pmc.json"), "w") as f:
        json.dump(pmc, f, indent=1, sort_keys=True)
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(tp) as f:
            traffic = json.load(f)
    except (OSError, ValueError):
        traffic = {}
    w = traffic.setdefault(workload, {})
    for k, cs in pmc.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            w[k] = {"hbm_bytes_per_launch": (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0,
                    "fetch_size_kib_raw": cs["FETCH_SIZE"], "write_size_kib": cs["WRITE_SIZE"], "source": tag}
    with open(tp, "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    for k, cs in sorted(pmc.items()):
        print(k, {c: round(v, 1) if isinstance(v, float) else v for c, v in cs.items() if not c.startswith("_")})


if __name__ == "__main__":
    main()
