#!/usr/bin/env python3
"""GPU box: seeded random frames at random sizes and thresholds, every stage against the CPU oracle
(the loop of tests/test_parity_gpu.py::test_random_frames_random_sizes over many more cases).
usage: python tools/soak_gpu.py [cases=80] [first_seed=5000]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    torch.cuda.init()
    import sift_amd as sift
    from oracle import oracle as orc
    import test_parity_gpu as T
    orc.build()
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    t0 = time.time()
    nk = 0
    for case in range(ncases):
        rng = np.random.default_rng(first + case)
        W = int(rng.integers(148, 1200))                              # any width, multiples of 4 or not
        H = int(rng.integers(145, 700))
        kind = case % 4
        if kind == 3:
            img = orc.synth_frame(first + case, W, H)
        else:
            yy, xx = np.mgrid[0:H, 0:W]
            img = np.zeros((H, W), np.float64)
            for _ in range(int(rng.integers(20, 300))):
                cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(1.5, 12)
                img += rng.uniform(-120, 120) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))
            img += rng.normal(0, rng.uniform(0, 8), img.shape)
            if kind == 0:
                img[:, W // 2:] += 90
            img = np.clip(img + 128, 0, 255).astype(np.uint8)
        th = int(rng.integers(1, 30))
        if case % 7 == 6:                                             # dense candidates: clusters in every 7x7 window, word edges included
            img = rng.integers(0, 256, img.shape).astype(np.uint8) if case % 2 else (img ^ rng.integers(0, 8, img.shape).astype(np.uint8))
            th = int(rng.integers(0, 3))
        _, _, n, total = T.check_full(sift, orc, img, th=th)
        nk += n
        if case % 3 == 0:                                             # the Cam1 geometry and the blur on the vector ALUs
            rk, rd, _ = orc.detect_and_compute(img, th, geometry=1)
            with sift.SiftContext(W, H, geometry=1) as ctx:
                ctx.set_blur_mfma(False)
                kps, desc = ctx.sift_detect_and_compute(img, th)
            assert (kps == rk).all() and (desc[:n] == rd[:n]).all() and not desc[n:].any(), "Cam1 case %d" % case
        if case % 10 == 9:
            print("case %d ok (%dx%d th %d: %d keypoints), %.0f s" % (case, W, H, th, n, time.time() - t0), flush=True)
    print("soak ok: %d cases, %d keypoints compared" % (ncases, nk))


if __name__ == "__main__":
    main()
