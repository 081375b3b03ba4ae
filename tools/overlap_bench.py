"""GPU box: do two batches on two streams overlap usefully?  Steps alternate between S contexts
(each its own stream and scratch); reports whole-job frames/s for S = 1, 2, 3.
usage: python tools/overlap_bench.py [--batch 64] [--steps 24]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sift_amd  # noqa: E402
from sift_amd.synth import synth_frame  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=24)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
W, H, B, K = a.width, a.height, a.batch, 10000
torch.cuda.init()
base = np.stack([synth_frame(1 + s, W, H) for s in range(8)])
for S in (1, 2, 3):
    per = B // S if S > 1 else B          # same number of frames in flight per "round"
    for frames_per_ctx in sorted({B, per}):
        imgs = np.concatenate([base] * ((frames_per_ctx + 7) // 8))[:frames_per_ctx]
        ctxs, bufs = [], []
        for _ in range(S):
            ctxs.append(sift_amd.SiftContext(W, H, max_batch=frames_per_ctx))
            bufs.append((torch.from_numpy(imgs).cuda(),
                         torch.zeros((frames_per_ctx, K, 2), dtype=torch.int32, device="cuda"),
                         torch.zeros((frames_per_ctx, K, 72), dtype=torch.int32, device="cuda")))
        def step(k):
            c, (di, dk, dd) = ctxs[k % S], bufs[k % S]
            c.detect_and_compute_dev(di, frames_per_ctx, dk, dd)
        for k in range(2 * S):
            step(k)
        for c in ctxs:
            c.sync()
        n = a.steps * S
        t0 = time.perf_counter()
        for k in range(n):
            step(k)
        for c in ctxs:
            c.sync()
        dt = time.perf_counter() - t0
        print("streams %d, %3d frames per call: %.0f frames/s" % (S, frames_per_ctx, n * frames_per_ctx / dt), flush=True)
        for c in ctxs:
            c.close()
        del bufs
        torch.cuda.empty_cache()
