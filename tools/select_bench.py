"""GPU box: time of the main.cpp:82-130 post-processing (sift_hip_select_top_dev) on a resident batch.
usage: python tools/select_bench.py [--batch 64] [--width 1920 --height 1080] [--nfinal 1000]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sift_amd  # noqa: E402
from sift_amd.synth import synth_frame  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--nfinal", type=int, default=1000)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
B, W, H, K, NF = a.batch, a.width, a.height, 10000, a.nfinal
imgs = np.stack([synth_frame(1 + s, W, H) for s in range(min(B, 8))])
imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
with sift_amd.SiftContext(W, H, max_batch=B) as ctx:
    d_img = torch.from_numpy(imgs).cuda()
    d_kps = torch.zeros((B, K, 2), dtype=torch.int32, device="cuda")
    d_desc = torch.zeros((B, K, 72), dtype=torch.int32, device="cuda")
    d_fin = torch.zeros((B, NF, 2), dtype=torch.int32, device="cuda")
    d_idx = torch.zeros((B, NF), dtype=torch.int32, device="cuda")
    d_nv = torch.zeros((B,), dtype=torch.int32, device="cuda")
    d_fdesc = torch.zeros((B, NF, 72), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ctx.detect_and_compute_dev(d_img, B, d_kps, d_desc, stream=st)
    res = {}
    for name, extra in (("select", (None, None)), ("select+descriptor rows", (d_desc, d_fdesc))):
        for _ in range(3):
            ctx.select_top_dev(d_kps, B, NF, d_fin, d_idx, d_nv, extra[0], extra[1], stream=st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ctx.select_top_dev(d_kps, B, NF, d_fin, d_idx, d_nv, extra[0], extra[1], stream=st)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / a.iters
    print(json.dumps({"workload": "%dx%d x%d, nfinal %d" % (W, H, B, NF), "ms_per_batch": res,
                      "nvalid": d_nv.cpu().tolist()[:4]}))
