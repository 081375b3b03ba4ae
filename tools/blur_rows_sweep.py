"""GPU box: k_blur time against the strip length (sift_hip_set_blur_rows) for small batches.
usage: python tools/blur_rows_sweep.py [--width 1920 --height 1080]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sift_amd  # noqa: E402
from sift_amd.synth import synth_frame  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
W, H, K = a.width, a.height, 10000
for B in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
    imgs = np.stack([synth_frame(1 + s, W, H) for s in range(B)])
    with sift_amd.SiftContext(W, H, max_batch=B) as ctx:
        d_img = torch.from_numpy(imgs).cuda()
        d_kps = torch.zeros((B, K, 2), dtype=torch.int32, device="cuda")
        d_desc = torch.zeros((B, K, 72), dtype=torch.int32, device="cuda")
        ctx.set_profiling(8)
        out = []
        for rows in (0,):
            ctx.set_blur_rows(rows)
            for _ in range(8):
                ctx.detect_and_compute_dev(d_img, B, d_kps, d_desc)
            ctx.sync()
            ms = [m["k_blur"] for m in ctx.stage_ms_all()][2:]
            out.append("%d:%.4f" % (rows, sum(ms) / len(ms)))
        print("batch %2d  k_blur ms by rows (0 = heuristic)  " % B + "  ".join(out), flush=True)
