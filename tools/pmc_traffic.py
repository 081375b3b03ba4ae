#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory into profiles/.

  python tools/pmc_traffic.py gpurun_out/prof_<tag> <tag> [--workload 1920x1080x256]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, our kernels),
profiles/<tag>_pmc.json (per-kernel mean of every counter collected in the separate --pmc
passes) and updates profiles/pmc_traffic.json (HBM bytes per launch, bench.py's roofline.traffic) and
profiles/pmc_units.json (what binds each kernel: VALU wave-instructions, matrix-core and texture-addresser
busy shares, the clock held during the kernel; bench.py's roofline_binding).  Both record the hash of the
kernel sources the profiled build was made from (written on the GPU box by tools/profile_gpu.sh);
bench.py quotes them only when the sources in the tree still have that hash.

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
    if not m:
        return None
    return "k_blur" if m.group(1) == "k_blur_mfma" else m.group(1)     # bench.py's stage name for either blur kernel


def main():
    src, tag = sys.argv[1], sys.argv[2]
    workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "1920x1080x256"
    with open(os.path.join(src, "kernel_sources.sha256")) as f:
        sha = f.read().strip()
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
    w = traffic[workload] = {"kernel_sources_sha256": sha}
    for k, cs in pmc.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            w[k] = {"hbm_bytes_per_launch": (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0,
                    "fetch_size_kib_raw": cs["FETCH_SIZE"], "write_size_kib": cs["WRITE_SIZE"], "source": tag}
    with open(tp, "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    # what binds each kernel.  GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles, so a launch lasts
    # GRBM_GUI_ACTIVE / 8 cycles; with its duration from the kernel trace that gives the clock held during it.
    dur = {}
    for p in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
        with open(p) as f:
            for r in csv.DictReader(f):
                k = short(r["Name"])
                if k:
                    dur[k] = float(r["AverageNs"]) * 1e-9
    up = os.path.join(ROOT, "profiles", "pmc_units.json")
    try:
        with open(up) as f:
            units = json.load(f)
    except (OSError, ValueError):
        units = {}
    uw = units[workload] = {"kernel_sources_sha256": sha, "source": tag,
                            "what": "per launch; *_frac = unit-busy cycles over the launch's cycles (GRBM_GUI_ACTIVE / 8 XCDs) "
                                    "x the number of units (1024 SIMDs, 256 texture addressers)"}
    clocks = []
    for k, cs in pmc.items():
        if "GRBM_GUI_ACTIVE" not in cs or "SQ_INSTS_VALU" not in cs:
            continue
        cyc = cs["GRBM_GUI_ACTIVE"] / 8.0
        e = {"SQ_INSTS_VALU": cs["SQ_INSTS_VALU"], "cycles_per_launch": cyc}
        if k in dur:
            e["avg_launch_s"] = dur[k]
            e["clock_hz"] = cyc / dur[k]
            clocks.append((dur[k], e["clock_hz"]))
        if "SQ_INSTS_MFMA" in cs:
            e["SQ_INSTS_MFMA"] = cs["SQ_INSTS_MFMA"]
        if cs.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            e["mfma_busy_frac"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        if "TA_TA_BUSY_sum" in cs:
            e["ta_busy_frac"] = cs["TA_TA_BUSY_sum"] / (256.0 * cyc)
        if "SQ_INSTS_LDS" in cs:
            e["SQ_INSTS_LDS"] = cs["SQ_INSTS_LDS"]
        if "SQ_LDS_IDX_ACTIVE" in cs:
            e["lds_active_frac"] = cs["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
        uw[k] = e
    if clocks:
        uw["sustained_clock_hz"] = sum(d * c for d, c in clocks) / sum(d for d, _ in clocks)
    with open(up, "w") as f:
        json.dump(units, f, indent=1, sort_keys=True)
    for k, cs in sorted(pmc.items()):
        print(k, {c: round(v, 1) if isinstance(v, float) else v for c, v in cs.items() if not c.startswith("_")})


if __name__ == "__main__":
    main()
