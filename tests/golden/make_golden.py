#!/usr/bin/env python3
"""Regenerates tests/golden/* (run in the build container; needs /root/reference for im0.jpg).

Fixtures are DATA only:
  im0_gray_1920x1080.png   the reference's test image src/data/im0.jpg decoded to 8-bit gray with
                           PIL convert('L') (SURVEY.md 8(d) cfg 1), stored losslessly so that every
                           consumer reads identical bytes
  im0_expected.npz         keypoints + descriptors of that image from the CPU oracle
  synth_expected.json      per (W,H,seed): keypoint count, uncapped total, pre-cull candidate count and
                           FNV-1a hashes of the oracle's keypoint array and valid descriptors
  survey_pins.json         facts SURVEY.md records from its survey-time run of the reference
                           (the only reference-derived anchors that exist; see oracle/sift_oracle.c)
The expected outputs come from the oracle, not from the reference (unbuildable here), so they pin
regressions of the oracle and serve as fast parity targets for the HIP path.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402


def fnv1a(buf):
    h = 0xCBF29CE484222325
    for b in np.frombuffer(buf, np.uint8).tolist():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def main():
    from PIL import Image
    src = "/root/reference/src/data/im0.jpg"
    png = os.path.join(HERE, "im0_gray_1920x1080.png")
    if os.path.exists(src):
        Image.open(src).convert("L").save(png, optimize=True)
    img = np.asarray(Image.open(png))
    kps, desc, total = orc.detect_and_compute(img)
    n = int((kps["contrast"] > 0).sum())
    np.savez_compressed(os.path.join(HERE, "im0_expected.npz"), kps=kps[:n], desc=desc[:n], total=total)
    exp = {}
    for (W, H, seed) in ((640, 480, 1), (640, 480, 2), (1920, 1080, 1), (1920, 1080, 2), (3840, 2160, 1), (256, 256, 3)):
        im = orc.synth_frame(seed, W, H)
        r = orc.sift_detect(im, planes=True, want_grad=False)
        k, d, tot = orc.detect_and_compute(im)
        nn = int((k["contrast"] > 0).sum())
        exp["%dx%d_s%d" % (W, H, seed)] = dict(n=nn, total=int(tot), candidates=int((r["kp1"] > 0).sum()),
                                               kps_fnv=fnv1a(k.tobytes()), desc_fnv=fnv1a(d[:nn].tobytes()))
        print(W, H, seed, exp["%dx%d_s%d" % (W, H, seed)], flush=True)
    with open(os.path.join(HERE, "synth_expected.json"), "w") as f:
        json.dump(exp, f, indent=1, sort_keys=True)
    pins = {
        "_source": "SURVEY.md section 8(c)/8(d)/Appendix A, recorded from the survey's run of the reference",
        "table_sums": [32765, 32759, 32744], "table_max": [9277, 1059, 304],
        "g1_centre_row": [265, 3814, 9277, 3814, 265],
        "g2_centre_row": [90, 218, 435, 713, 959, 1059, 959, 713, 435, 218, 90],
        "g3_centre_row": [32, 52, 78, 112, 152, 195, 237, 272, 295, 304, 295, 272, 237, 195, 152, 112, 78, 52, 32],
        "circle_mask_popcount": 517, "mask_all_overlap_pixels": 844, "mask_all_pixels": 3809,
        "keypoints": {"im0": 218, "640x480_s1": 985, "1920x1080_s1": 9493, "3840x2160_s1": 10000},
        "candidates": {"im0": 312, "1920x1080_s1": 13328},
    }
    with open(os.path.join(HERE, "survey_pins.json"), "w") as f:
        json.dump(pins, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
