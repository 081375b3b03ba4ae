#!/usr/bin/env python3
"""PCIe-inclusive throughput of sift_hip_batch (pinned host frames in, pinned host results out).
Not bench.py's `value`: DESIGN.md quotes this number next to the device-resident one."""
import ctypes as C
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import sift_amd
from sift_amd.hw_sift import batch


def pinned(lib, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = lib.sift_hip_host_alloc(n)
    assert p, "pinned allocation failed"
    buf = (C.c_char * n).from_address(p)
    return np.frombuffer(buf, dtype=dtype).reshape(shape), p


def main():
    W, H = 1920, 1080
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    max_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    lib = sift_amd.load_library()
    K = sift_amd.MAX_KEYPOINTS
    base = [sift_amd.synth_frame(1 + i, W, H) for i in range(16)]
    imgs, pi = pinned(lib, (nframes, H, W), np.uint8)
    for i in range(nframes):
        imgs[i] = base[i % 16]
    kps, pk = pinned(lib, (nframes, K), sift_amd.KP_DTYPE)
    desc, pd = pinned(lib, (nframes, K, 72), np.uint32)
    ctx = sift_amd.SiftContext(W, H, max_batch=max_batch)
    hs = (C.c_void_p * 1)(ctx._h)
    ps = (C.c_void_p * nframes)(*[imgs[i].ctypes.data for i in range(nframes)])
    vp = lambda a: C.c_void_p(a.ctypes.data)
    for rep in range(3):
        t = time.perf_counter()
        sift_amd.capi.check(lib.sift_hip_batch(hs, 1, ps, nframes, vp(kps), vp(desc), 15))
        dt = time.perf_counter() - t
        print("sift_hip_batch: %d frames, max_batch %d: %.1f ms -> %.0f frames/s (PCIe in + out included), kp[0]=%d"
              % (nframes, max_batch, dt * 1e3, nframes / dt, int((kps[0]["contrast"] > 0).sum())), flush=True)
    NF = 1000
    fin, pf = pinned(lib, (nframes, NF), sift_amd.KP_DTYPE)
    idx, px = pinned(lib, (nframes, NF), np.uint32)
    fdesc, pfd = pinned(lib, (nframes, NF, 72), np.uint32)
    nv, pn = pinned(lib, (nframes,), np.uint32)
    for rep in range(3):
        t = time.perf_counter()
        sift_amd.capi.check(lib.sift_hip_batch_top(hs, 1, ps, nframes, NF, vp(fin), vp(idx), vp(fdesc), vp(nv), 15))
        dt = time.perf_counter() - t
        print("sift_hip_batch_top (1000 strongest keypoints + their descriptor rows back): %d frames: %.1f ms -> %.0f frames/s, nKp[0]=%d"
              % (nframes, dt * 1e3, nframes / dt, int(nv[0])), flush=True)
    ctx.close()
    for p in (pf, px, pfd, pn):
        lib.sift_hip_host_free(p)
    for p in (pi, pk, pd):
        lib.sift_hip_host_free(p)


if __name__ == "__main__":
    main()
