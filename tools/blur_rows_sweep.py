#!/usr/bin/env python3
"""GPU box: k_blur time per 256-frame launch for several strip lengths (sift_hip_set_blur_rows).
usage: python tools/blur_rows_sweep.py [rows ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import sift_amd as sift
    from sift_amd.synth import synth_frames_torch
    W, H, B = 1920, 1080, 256
    dev = torch.device("cuda:0")
    rows = [int(a) for a in sys.argv[1:]] or [0, 1080, 540, 360, 270, 216, 180, 135]
    rows = rows + rows                                            # twice round: clocks and caches settle during the first
    d_img = synth_frames_torch(range(1, B + 1), W, H, dev)
    with sift.SiftContext(W, H, max_batch=B) as ctx:
        K = ctx.max_keypoints
        d_k = torch.zeros((B, K, 2), dtype=torch.int32, device=dev)
        d_d = torch.zeros((B, K, 72), dtype=torch.int32, device=dev)
        ctx.set_profiling(24)
        for r in rows:
            ctx.set_blur_rows(r)
            for _ in range(3):
                ctx.detect_and_compute_dev(d_img.data_ptr(), B, d_k.data_ptr(), d_d.data_ptr())
            ctx.sync()
            for _ in range(24):
                ctx.detect_and_compute_dev(d_img.data_ptr(), B, d_k.data_ptr(), d_d.data_ptr())
            ctx.sync()
            all_ms = ctx.stage_ms_all()
            print("blur_rows %4d: k_blur %.4f ms (mean of %d calls), k_detect %.4f" % (r, np.mean([m["k_blur"] for m in all_ms]), len(all_ms), np.mean([m["k_detect"] for m in all_ms])), flush=True)


if __name__ == "__main__":
    main()
