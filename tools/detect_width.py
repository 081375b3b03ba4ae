#!/usr/bin/env python3
"""GPU box: does the half-empty eighth wave of k_detect at 1080p cost time?  A row of 1920 pixels is 480 four-pixel
groups = 7.5 waves; 1792 is 7 full waves and 2048 is 8.  If the wave count set the time, 1920 would cost what 2048
costs; if the bytes do, time follows the width.  Prints every kernel's time per 256-frame launch at the three
widths and the time per megapixel.
usage: python tools/detect_width.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import sift_amd as sift
    from sift_amd.synth import synth_frames_torch
    H, B = 1080, 256
    dev = torch.device("cuda:0")
    for rep in range(2):
        for W in (1792, 1920, 2048):
            d_img = synth_frames_torch(range(1, B + 1), W, H, dev)
            with sift.SiftContext(W, H, max_batch=B) as ctx:
                K = ctx.max_keypoints
                d_k = torch.zeros((B, K, 2), dtype=torch.int32, device=dev)
                d_d = torch.zeros((B, K, 72), dtype=torch.int32, device=dev)
                ctx.set_profiling(16)
                for _ in range(3):
                    ctx.detect_and_compute_dev(d_img.data_ptr(), B, d_k.data_ptr(), d_d.data_ptr())
                ctx.sync()
                for _ in range(16):
                    ctx.detect_and_compute_dev(d_img.data_ptr(), B, d_k.data_ptr(), d_d.data_ptr())
                ctx.sync()
                ms = ctx.stage_ms_all()
                mp = W * H * B / 1e6
                t = {k: float(np.mean([m[k] for m in ms])) for k in ("k_blur", "k_detect", "k_cull", "k_emit", "k_describe")}
                print("W %4d (%.2f waves per row): " % (W, W / 256.0) +
                      "  ".join("%s %.4f ms (%.3f us/Mpx)" % (k, v, 1e3 * v / mp) for k, v in t.items()), flush=True)
            del d_img, d_k, d_d
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
