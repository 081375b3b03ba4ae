#!/usr/bin/env python3
"""GPU box: how fast one host thread zero-fills pinned memory (memset through numpy) -- the cost a valid-rows-only D2H
copy of sift_hip_batch would move to the host for sparse frames.  usage: python tools/host_fill_rate.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import sift_amd
    lib = sift_amd.load_library()
    n = 128 * 10000 * 72 * 4                                   # the descriptor matrix of 128 frames
    p = lib.sift_hip_host_alloc(n)
    a = np.frombuffer((C.c_uint8 * n).from_address(p), dtype=np.uint8)
    a[:] = 1
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        C.memset(p, 0, n)
        best = min(best, time.perf_counter() - t0)
    per_frame = 10000 * 72 * 4 - 218 * 72 * 4
    print("one thread zero-fills pinned memory at %.1f GB/s: the %.2f MB of padding rows of a 218-keypoint frame take %.0f us -> %.0f frames/s per thread"
          % (n / best / 1e9, per_frame / 1e6, per_frame / (n / best) * 1e6, (n / best) / per_frame))
    lib.sift_hip_host_free(p)


if __name__ == "__main__":
    main()
