#!/usr/bin/env python3
"""GPU box: the three stand-alone entry points (HW_SIFT.cpp:29-41 called one by one, Grad_t planes
in HBM between them) against the fused call, device-resident, 1080p.

  python tools/standalone_bench.py [frames=16] [reps=10]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    torch.cuda.init()
    from sift_amd import hw_sift as hs
    from oracle import oracle as orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    W, H, K = 1920, 1080, 10000
    dev = torch.device("cuda:0")
    imgs = torch.from_numpy(np.stack([orc.synth_frame(s + 1, W, H) for s in range(n)])).to(dev)
    grad = torch.empty((n, H, W), dtype=torch.int32, device=dev)
    kps = torch.empty((n, K, 2), dtype=torch.int32, device=dev)
    kps2 = torch.empty_like(kps)
    ori = torch.empty((n, K), dtype=torch.uint8, device=dev)
    desc = torch.empty((n, K, 72), dtype=torch.int32, device=dev)
    desc2 = torch.empty_like(desc)
    with hs.SiftContext(W, H, max_batch=n) as ctx:
        def t(f):
            f(); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            ctx.sync()
            return (time.perf_counter() - t0) / reps * 1e3
        p = lambda x: x.data_ptr()
        fused = t(lambda: ctx.detect_and_compute_dev(p(imgs), n, p(kps2), p(desc2)))
        d = t(lambda: ctx.detect_dev(p(imgs), n, p(grad), p(kps)))
        o = t(lambda: ctx.main_orientation_dev(p(grad), n, p(kps), p(ori)))
        c = t(lambda: ctx.descriptor_dev(p(grad), n, p(kps), p(ori), p(desc)))
        same = bool((kps == kps2).all() and (desc == desc2).all())
        print("frames %d: fused %.3f ms | sift_detect %.3f + compute_main_orientation %.3f + compute_descriptor %.3f = %.3f ms | same outputs: %s"
              % (n, fused, d, o, c, d + o + c, same))
        print("frames/s: fused %.0f, three calls %.0f" % (n / fused * 1e3, n / (d + o + c) * 1e3))


if __name__ == "__main__":
    main()
