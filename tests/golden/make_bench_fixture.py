#!/usr/bin/env python3
"""Per-seed expectations for bench.py's post-run check, made by the CPU oracle.

  python tests/golden/make_bench_fixture.py            -> tests/golden/bench_expected.json

For every synthetic frame bench.py can run (1920x1080 seeds 1..2048 = 8 ranks x 256 frames,
640x480 seeds 1..256, 3840x2160 seeds 1..64; contrastTh 15) the oracle's uncapped survivor count
and an order-sensitive 64-bit checksum of the keypoint array (10000 slots) and of the descriptor
matrix (10000 x 72 words, padding rows zero).  bench.py computes the same checksums of its device
buffers after the timed loop (sift_amd/checksum.py) and compares them frame by frame.
"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc            # noqa: E402
from sift_amd.checksum import words_checksum_np   # noqa: E402

CONFIGS = (("1920x1080", 1920, 1080, 2048), ("640x480", 640, 480, 256), ("3840x2160", 3840, 2160, 64))


def one(args):
    W, H, seed = args
    kps, desc, total = orc.detect_and_compute(orc.synth_frame(seed, W, H), 15)
    n = int((kps["contrast"] > 0).sum())
    desc[n:] = 0                                      # padding rows are defined as 0
    return [int(total), "%016x" % words_checksum_np(kps.view(np.uint32)), "%016x" % words_checksum_np(desc)]


def main():
    orc.build()
    out = {"contrast_th": 15, "what": "[total, keypoint checksum, descriptor checksum] per seed, seeds from 1"}
    workers = len(os.sched_getaffinity(0))
    for name, W, H, n in CONFIGS:
        with ThreadPoolExecutor(workers) as ex:
            out[name] = list(ex.map(one, [(W, H, s) for s in range(1, n + 1)]))
        print(name, n, "frames", file=sys.stderr)
    with open(os.path.join(ROOT, "tests", "golden", "bench_expected.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
