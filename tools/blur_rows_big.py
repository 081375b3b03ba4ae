"""GPU box: k_blur time per 64 frames against the strip length for large batches.
usage: python tools/blur_rows_big.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sift_amd  # noqa: E402
from sift_amd.synth import synth_frame  # noqa: E402

W, H, K = 1920, 1080, 10000
base = np.stack([synth_frame(1 + s, W, H) for s in range(16)])
for B in (64, 96, 128, 192, 256, 384):
    imgs = np.concatenate([base] * (B // 16))
    with sift_amd.SiftContext(W, H, max_batch=B) as ctx:
        d_img = torch.from_numpy(imgs).cuda()
        d_kps = torch.zeros((B, K, 2), dtype=torch.int32, device="cuda")
        d_desc = torch.zeros((B, K, 72), dtype=torch.int32, device="cuda")
        ctx.set_profiling(8)
        out = []
        for rows in (0, 1080, 540, 360, 270, 216, 180, 135):
            ctx.set_blur_rows(rows)
            for _ in range(6):
                ctx.detect_and_compute_dev(d_img, B, d_kps, d_desc)
            ctx.sync()
            ms = [m["k_blur"] for m in ctx.stage_ms_all()][2:]
            out.append("%d:%.4f" % (rows, sum(ms) / len(ms) * 64 / B))
        print("batch %3d  k_blur ms per 64 frames by rows (0 = heuristic)  " % B + "  ".join(out), flush=True)
    del d_img, d_kps, d_desc
    torch.cuda.empty_cache()
