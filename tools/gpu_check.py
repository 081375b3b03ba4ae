#!/usr/bin/env python3
"""Quick stage-by-stage parity report HIP vs oracle on one synthetic frame (GPU box)."""
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import sift_amd
from oracle import oracle as orc


def main():
    W, H, seed = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480, 1)
    img = sift_amd.synth_frame(seed, W, H)
    t = time.time()
    ref = orc.sift_detect(img, planes=True)
    rk, rd, _ = orc.detect_and_compute(img)
    print("oracle %.2fs, kp=%d" % (time.time() - t, int((rk["contrast"] > 0).sum())), flush=True)
    ctx = sift_amd.SiftContext(W, H)
    t = time.time()
    kps, desc = ctx.sift_detect_and_compute(img)
    print("hip %.3fs" % (time.time() - t), flush=True)
    pl = ctx.debug_planes(0)
    for n in ("g1", "g2", "g3"):
        d = pl[n] != ref[n]
        print(n, "mismatch", int(d.sum()), "first", np.argwhere(d)[:3].tolist())
    d = pl["cand"] != ref["kp1"]
    print("cand mismatch", int(d.sum()), np.argwhere(d)[:3].tolist())
    m = ref["kp1"] > 0
    d = (pl["eratio"] != ref["eratio"]) & m
    print("eratio mismatch at candidates", int(d.sum()))
    d = pl["surv"] != ref["kp2"]
    print("surv mismatch", int(d.sum()), np.argwhere(d)[:3].tolist())
    print("kps equal", bool((kps == rk).all()), "n", int((kps["contrast"] > 0).sum()))
    n = int((rk["contrast"] > 0).sum())
    print("desc equal (valid)", bool((desc[:n] == rd[:n]).all()), "padding zero", bool((desc[n:] == 0).all()))
    if not (desc[:n] == rd[:n]).all():
        bad = np.argwhere(desc[:n] != rd[:n])
        print("desc first bad", bad[:5].tolist(), desc[bad[0][0]][:16], rd[bad[0][0]][:16])
    grad, kps2 = ctx.sift_detect(img)
    g = ref["grad"]
    d = (grad["ang"] != g["ang"]) | (grad["mag"] != g["mag"])
    print("grad mismatch", int(d.sum()), np.argwhere(d)[:3].tolist(), "kps2 equal", bool((kps2 == rk).all()))
    ori = ctx.compute_main_orientation(g, rk)
    ro = orc.main_orientation(g, rk)
    print("ori equal", bool((ori == ro).all()))
    d2 = ctx.compute_descriptor(g, rk, ro)
    print("desc(api) equal", bool((d2[:n] == rd[:n]).all()))
    ctx.set_profiling(True)
    for _ in range(3):
        ctx.sift_detect_and_compute(img)
    print(ctx.stage_ms())


if __name__ == "__main__":
    main()
