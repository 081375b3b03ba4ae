#!/usr/bin/env python3
"""GPU box: k_detect time per 256-frame launch against its two launch parameters, the LDS queue length at which
it drains (sift_hip_set_detect_queue) and the strip length (sift_hip_set_detect_rows).
usage: python tools/detect_sweep.py"""
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
    d_img = synth_frames_torch(range(1, B + 1), W, H, dev)
    with sift.SiftContext(W, H, max_batch=B) as ctx:
        K = ctx.max_keypoints
        d_k = torch.zeros((B, K, 2), dtype=torch.int32, device=dev)
        d_d = torch.zeros((B, K, 72), dtype=torch.int32, device=dev)
        ctx.set_profiling(16)
        for rep in range(2):
            for rows in (32, 24, 16, 12, 8):
                for q in (0,):
                    ctx.set_detect_rows(rows)
                    ctx.set_detect_queue(q)
                    for _ in range(3):
                        ctx.detect_and_compute_dev(d_img.data_ptr(), B, d_k.data_ptr(), d_d.data_ptr())
                    ctx.sync()
                    for _ in range(16):
                        ctx.detect_and_compute_dev(d_img.data_ptr(), B, d_k.data_ptr(), d_d.data_ptr())
                    ctx.sync()
                    ms = ctx.stage_ms_all()
                    print("rows %2d queue %4d: k_detect %.4f ms  k_cull %.4f" % (rows, q, np.mean([m["k_detect"] for m in ms]), np.mean([m["k_cull"] for m in ms])), flush=True)


if __name__ == "__main__":
    main()
