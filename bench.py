#!/usr/bin/env python3
"""bench.py -- 1080p frames/s of the full SIFT hot path (keypoints + descriptors) on MI355X.

A step = one pass of sift_detect_and_compute (HW_SIFT.cpp:16-50) over one batch of synthetic
1920x1080 frames that is already resident in HBM; outputs stay in HBM.  One process per GPU;
frames are independent, so for N > 1 there is no data-path collective: every rank processes its own
batch (--scaling weak, the default) or its contiguous share of one batch (--scaling strong,
SURVEY.md 8(e)), and the job time is the max over ranks.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python bench.py --gpus N ...                 (starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").  After the timed loop every frame's
outputs are compared with the oracle's (tests/golden/bench_expected.json: per-seed totals and
checksums of the keypoint array and the descriptor matrix); a mismatch ends the run non-zero.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s
I8_MFMA_PEAK_TOPS = 5000.0                  # MI355X_MICROARCH.md, matrix cores: I8 = 2x BF16 = ~5 POP/s dense
NSIMD = 256 * 4
NCU = 256
PEAK_CLOCK_HZ = 2.4e9


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256,
                    help="frames per step: per GPU (weak) or in total (strong); the kernels reach their steady rate from ~200 frames per GPU on")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--contrast-th", type=int, default=15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the comparison with the oracle's fixture (kernel experiments)")
    ap.add_argument("--overlap", action="store_true",
                    help="also time the steps alternating between two contexts (two streams); level with or slower than one "
                         "stream since the blur runs on the matrix cores, so not part of the default run")
    ap.add_argument("--no-overlap", action="store_true", help="(default; kept for older scripts)")
    ap.add_argument("--no-profile-events", action="store_true", help="leave the per-kernel HIP events out")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the 640x480 and 3840x2160 passes")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the pinned-host-buffer passes")
    return ap.parse_args(argv)


def self_launch(a, argv):
    """--gpus N > 1 without an outer launcher: start the N ranks as children (torch.distributed.run),
    before this process imports torch or touches a GPU; their rank 0 prints the JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def cpu_baseline(width, height, budget_s=12.0, max_frames=12):
    """The CPU oracle (a port of the reference's algorithm, oracle/sift_oracle.c) on one host core,
    on a bounded sample of the same workload (seeds 1..n, the first frames of rank 0's batch).
    Returns the record and the oracle's outputs, which the caller compares with the GPU's."""
    from oracle import oracle as orc
    orc.build()
    frames = [orc.synth_frame(1 + i, width, height) for i in range(max_frames)]
    orc.detect_and_compute(frames[0])       # page in
    res, t0 = [], time.perf_counter()
    while len(res) < max_frames and (len(res) < 3 or time.perf_counter() - t0 < budget_s):
        res.append(orc.detect_and_compute(frames[len(res)]))
    dt = time.perf_counter() - t0
    n = len(res)
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d synthetic %dx%d frames (seeds 1..%d), full pipeline, %.1f s" % (n, width, height, n, dt)}, res


def cpu_baseline_parallel(width, height, frames_per_thread=2):
    """The same oracle, one frame per call, frame-parallel over the host cores this process may use
    (the reference is single-threaded; frames are independent).  ctypes releases the
    GIL inside the C call, so plain threads run in parallel."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)), 64))
    frames = [orc.synth_frame(1 + i % 12, width, height) for i in range(cores * frames_per_thread)]
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(orc.detect_and_compute, frames))
        dt = time.perf_counter() - t0
    return {"value": len(frames) / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d synthetic %dx%d frames over %d threads, full pipeline, %.1f s" % (len(frames), width, height, cores, dt)}


def kernel_sources_sha256():
    """hash of the kernel sources in the tree: the committed counter summaries name the hash of the
    build they were measured on, and are only quoted when it is this one"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sift_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def load_counters(workload):
    """Per-kernel counter summaries of `workload` from the committed PMC passes (profiles/pmc_traffic.json
    and profiles/pmc_units.json, written by tools/pmc_traffic.py from separate rocprofv3 --pmc runs), or {}
    when they were measured on other kernel sources than the ones in the tree."""
    out = {}
    sha = kernel_sources_sha256()
    for name in ("pmc_traffic.json", "pmc_units.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        e = d.get(workload)
        if isinstance(e, dict) and e.get("kernel_sources_sha256") == sha:
            out[name] = e
    return out


def expected_for(W, H, th):
    """the oracle's per-seed [total, keypoint checksum, descriptor checksum] for this configuration, or None"""
    if th != 15:
        return None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "bench_expected.json")) as f:
            return json.load(f).get("%dx%d" % (W, H))
    except (OSError, ValueError):
        return None


def check_outputs(seeds, d_kps, d_desc, totals, expected, what):
    """every frame's total and checksums against the fixture -> number of frames checked"""
    from sift_amd.checksum import words_checksum_torch
    if expected is None or not seeds or max(seeds) > len(expected):
        return 0
    ksum = words_checksum_torch(d_kps)
    dsum = words_checksum_torch(d_desc)
    for i, s in enumerate(seeds):
        tot, kx, dx = expected[s - 1]
        if int(totals[i]) != tot or "%016x" % ksum[i] != kx or "%016x" % dsum[i] != dx:
            raise SystemExit("wrong result (%s): seed %d gives total %d, keypoints %016x, descriptors %016x; the oracle's are %d, %s, %s"
                             % (what, s, int(totals[i]), ksum[i], dsum[i], tot, kx, dx))
    return len(seeds)


def run_config(sift_amd, torch, dev, W, H, seeds, steps, warmup, th, spin_s=0.5, profile=True, d_img=None):
    """frames of `seeds` (or the frames in d_img) resident on GPU `dev`, a context, spin-up and warm-up steps done ->
    (context, the step function, (d_img, d_kps, d_desc, d_tot))"""
    from sift_amd.synth import synth_frames_torch
    B, K = len(seeds), sift_amd.MAX_KEYPOINTS
    device = "cuda:%d" % dev
    if d_img is None:
        d_img = torch.cat([synth_frames_torch(seeds[i:i + 32], W, H, device) for i in range(0, B, 32)])
    d_kps = torch.zeros((B, K, 2), dtype=torch.int32, device=device)
    d_desc = torch.zeros((B, K, 72), dtype=torch.int32, device=device)
    d_tot = torch.zeros(B, dtype=torch.int32, device=device)
    torch.cuda.synchronize()
    ctx = sift_amd.SiftContext(W, H, device=dev, max_batch=B)

    def step():
        ctx.detect_and_compute_dev(d_img, B, d_kps, d_desc, d_tot, th)

    # spin-up, before the warm-up steps: the GPU's clocks take tens of milliseconds of load to
    # settle (the first steps after an idle period run several per cent slower)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < spin_s:
        step()
        ctx.sync()
    for _ in range(warmup):
        step()
    ctx.sync()
    if profile:
        ctx.set_profiling(max(1, min(steps, 4096)))
    return ctx, step, (d_img, d_kps, d_desc, d_tot)


def stage_summary(sift_amd, ctx):
    recs = ctx.stage_ms_all()
    stage = {k: float(np.mean([r[k] for r in recs])) for k in sift_amd.STAGE_NAMES}
    per_step = np.array([sum(r[k] for k in sift_amd.STAGE_NAMES) for r in recs])
    spread = {"p5": float(np.percentile(per_step, 5)), "p50": float(np.percentile(per_step, 50)),
              "p95": float(np.percentile(per_step, 95)), "steps": int(len(per_step)),
              "what": "sum of the five kernels' HIP-event times per step, ms"}
    return stage, spread


def other_config(sift_amd, torch, dev, W, H, B, th, steps=10, warmup=2):
    """a BASELINE.json configuration other than the headline one, device-resident, on this GPU"""
    seeds = list(range(1, B + 1))
    ctx, step, (d_img, d_kps, d_desc, d_tot) = run_config(sift_amd, torch, dev, W, H, seeds, steps, warmup, th, spin_s=0.1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    el = time.perf_counter() - t0
    stage, _ = stage_summary(sift_amd, ctx)
    totals = d_tot.cpu().numpy()
    checked = check_outputs(seeds, d_kps, d_desc, totals, expected_for(W, H, th), "%dx%d" % (W, H))
    ctx.close()
    return {"workload": "%dx%d synthetic frames, full pipeline" % (W, H), "frames_per_step": B, "steps": steps,
            "value": B * steps / el, "unit": "frames/s", "ms_per_step": 1e3 * el / steps,
            "keypoints_per_frame_mean": float(np.minimum(totals, sift_amd.MAX_KEYPOINTS).mean()),
            "kernel_ms_per_step": stage, "checked_frames": checked}


def im0_config(sift_amd, torch, dev, th, B=256, steps=10, warmup=2, with_cpu=True):
    """BASELINE.json configs[0]: the reference's test image (src/data/im0.jpg as the committed 8-bit gray
    fixture, 218 keypoints: the sparse regime), a batch of B copies of it device-resident, and the oracle on it
    on one host core.  Every frame's outputs are compared with tests/golden/im0_expected.npz."""
    from PIL import Image
    g = os.path.join(ROOT, "tests", "golden")
    im0 = np.ascontiguousarray(np.asarray(Image.open(os.path.join(g, "im0_gray_1920x1080.png")), dtype=np.uint8))
    z = np.load(os.path.join(g, "im0_expected.npz"))
    H, W = im0.shape
    K, n = sift_amd.MAX_KEYPOINTS, len(z["kps"])
    d_img = torch.from_numpy(im0).to("cuda:%d" % dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
    ctx, step, (d_img, d_kps, d_desc, d_tot) = run_config(sift_amd, torch, dev, W, H, list(range(B)), steps, warmup, th,
                                                            spin_s=0.1, d_img=d_img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    el = time.perf_counter() - t0
    stage, _ = stage_summary(sift_amd, ctx)
    ctx.close()
    checked = 0
    if th == 15:
        gk = d_kps[0].cpu().numpy().view(sift_amd.KP_DTYPE).reshape(K)
        gd = d_desc[0].cpu().numpy().view(np.uint32)
        same = bool((d_kps == d_kps[0]).all() and (d_desc == d_desc[0]).all() and (d_tot == n).all())
        if not ((gk[:n] == z["kps"]).all() and not gk[n:].view(np.uint64).any() and (gd[:n] == z["desc"]).all()
                and not gd[n:].any() and same):
            raise SystemExit("wrong result (im0): the frames differ from tests/golden/im0_expected.npz")
        checked = B
    out = {"workload": "reference test image im0 (1920x1080 gray, configs[0]), full pipeline", "frames_per_step": B,
           "steps": steps, "value": B * steps / el, "unit": "frames/s", "ms_per_step": 1e3 * el / steps,
           "keypoints_per_frame_mean": float(n), "kernel_ms_per_step": stage, "checked_frames": checked}
    if with_cpu:
        from oracle import oracle as orc
        orc.detect_and_compute(im0, th)
        t0, reps = time.perf_counter(), 3
        for _ in range(reps):
            rk, rd, total = orc.detect_and_compute(im0, th)
        dt = time.perf_counter() - t0
        if th == 15 and not (total == n and (rk[:n] == z["kps"]).all() and (rd[:n] == z["desc"]).all()):
            raise SystemExit("the oracle's im0 outputs differ from tests/golden/im0_expected.npz")
        out["cpu_baseline"] = {"value": reps / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d passes over im0, full pipeline, %.1f s" % (reps, dt)}
    return out


def octave_config(sift_amd, torch, dev, W, H, B, th, steps=6, warmup=2, with_cpu=True, oracle_frames=2):
    """BASELINE.json configs[3], "4K frames, extra octave": the reference has one octave, so the
    second one is this build's extension (include/sift_hip.h, "further octaves": pixel (2x, 2y) of G2 through the
    unchanged pipeline at half size) -- PARITY UNPINNED.  A step = octave 0 of B frames, the decimation, octave 1
    of the B half-size frames, all on one stream, device-resident.  Octave 0 is checked against the oracle's
    fixture, both octaves of the first frames against the oracle's composition run here."""
    from sift_amd.synth import synth_frames_torch
    K, device = sift_amd.MAX_KEYPOINTS, "cuda:%d" % dev
    seeds = list(range(1, B + 1))
    W1, H1 = W // 2, H // 2
    d_img = torch.cat([synth_frames_torch(seeds[i:i + 16], W, H, device) for i in range(0, B, 16)])
    d_img1 = torch.zeros((B, H1, W1), dtype=torch.uint8, device=device)
    outs = [(torch.zeros((B, K, 2), dtype=torch.int32, device=device), torch.zeros((B, K, 72), dtype=torch.int32, device=device),
             torch.zeros(B, dtype=torch.int32, device=device)) for _ in range(2)]
    c0 = sift_amd.SiftContext(W, H, device=dev, max_batch=B)
    c1 = sift_amd.SiftContext(W1, H1, device=dev, max_batch=B)
    stream = torch.cuda.Stream(device=device)
    sp = stream.cuda_stream

    def step():
        c0.detect_and_compute_dev(d_img, B, outs[0][0], outs[0][1], outs[0][2], th, stream=sp)
        c0.next_octave_dev(0, B, d_img1, plane=2, stream=sp)
        c1.detect_and_compute_dev(d_img1, B, outs[1][0], outs[1][1], outs[1][2], th, stream=sp)

    for _ in range(warmup):
        step()
    stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    stream.synchronize()
    el = time.perf_counter() - t0
    tot0, tot1 = outs[0][2].cpu().numpy(), outs[1][2].cpu().numpy()
    checked = check_outputs(seeds, outs[0][0], outs[0][1], tot0, expected_for(W, H, th), "%dx%d, octave 0" % (W, H))
    out = {"workload": "%dx%d synthetic frames + one further octave (%dx%d), full pipeline per octave (configs[3]); the second "
                       "octave is an extension, parity unpinned" % (W, H, W1, H1),
           "frames_per_step": B, "steps": steps, "value": B * steps / el, "unit": "frames/s", "ms_per_step": 1e3 * el / steps,
           "keypoints_per_frame_mean": [float(np.minimum(tot0, K).mean()), float(np.minimum(tot1, K).mean())],
           "checked_frames": checked}
    if with_cpu:
        from oracle import oracle as orc
        t0 = time.perf_counter()
        for i in range(oracle_frames):
            want = orc.detect_and_compute_octaves(orc.synth_frame(seeds[i], W, H), 2, th)
            for o, (rk, rd, total) in enumerate(want):
                n = int((rk["contrast"] > 0).sum())
                gk = outs[o][0][i].cpu().numpy().view(sift_amd.KP_DTYPE).reshape(K)
                gd = outs[o][1][i].cpu().numpy().view(np.uint32)
                if not ((gk == rk).all() and (gd[:n] == rd[:n]).all() and not gd[n:].any() and int((tot0, tot1)[o][i]) == total):
                    raise SystemExit("wrong result (%dx%d + octave): octave %d of seed %d differs from the oracle's composition" % (W, H, o, seeds[i]))
        dt = time.perf_counter() - t0
        out["checked_frames_both_octaves"] = oracle_frames
        out["cpu_baseline"] = {"value": oracle_frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d synthetic %dx%d frames, both octaves, %.1f s" % (oracle_frames, W, H, dt)}
    c0.close(); c1.close()
    return out


def end_to_end(sift_amd, dev, W, H, th, expected, frames_per_call=128, calls=8, max_batch=64, nfinal=1000):
    """SURVEY.md 8(d) timing (b): frames in pinned host memory, results back in pinned host memory
    (the bracket of main.cpp:55-77 around the image loop), through sift_hip_batch (every keypoint slot and
    descriptor row returns) and sift_hip_batch_top (the selection of main.cpp:82-130 on the device, the
    1000 strongest keypoints and their descriptor rows return); one call = frames_per_call frames,
    double-buffered H2D / kernels / D2H inside the call.  PCIe-inclusive: never the headline value."""
    import ctypes as C
    lib = sift_amd.load_library()
    K, NF, n, np_ = sift_amd.MAX_KEYPOINTS, nfinal, frames_per_call, W * H

    def pinned(nbytes, dtype, shape):
        p = lib.sift_hip_host_alloc(nbytes)
        if not p:
            raise SystemExit("sift_hip_host_alloc(%d) failed" % nbytes)
        return p, np.frombuffer((C.c_uint8 * nbytes).from_address(p), dtype=dtype).reshape(shape)

    bufs = []
    try:
        p_img, imgs = pinned(n * np_, np.uint8, (n, H, W)); bufs.append(p_img)
        p_kps, kps = pinned(n * K * 8, np.uint32, (n, K * 2)); bufs.append(p_kps)
        p_desc, desc = pinned(n * K * 72 * 4, np.uint32, (n, K * 72)); bufs.append(p_desc)
        p_fk, fk = pinned(n * NF * 8, np.uint32, (n, NF * 2)); bufs.append(p_fk)
        p_fi, fi = pinned(n * NF * 4, np.uint32, (n, NF)); bufs.append(p_fi)
        p_fd, fd = pinned(n * NF * 72 * 4, np.uint32, (n, NF * 72)); bufs.append(p_fd)
        p_nv, nv = pinned(n * 4, np.uint32, (n,)); bufs.append(p_nv)
        seeds = list(range(1, n + 1))
        for i, s in enumerate(seeds):
            imgs[i] = sift_amd.synth_frame(s, W, H)
        ctx = sift_amd.SiftContext(W, H, device=dev, max_batch=max_batch)
        hs = (C.c_void_p * 1)(ctx._h)
        ps = (C.c_void_p * n)(*[p_img + i * np_ for i in range(n)])

        def timed(call):
            for _ in range(2):
                sift_amd.capi.check(call())
            ms = []
            for _ in range(calls):
                t0 = time.perf_counter()
                sift_amd.capi.check(call())
                ms.append(1e3 * (time.perf_counter() - t0))
            ms = np.array(ms)
            return {"frames_per_call": n, "calls": calls, "value": n * calls / (ms.sum() * 1e-3), "unit": "frames/s",
                    "ms_per_call": {"p5": float(np.percentile(ms, 5)), "p50": float(np.percentile(ms, 50)),
                                    "p95": float(np.percentile(ms, 95))}}

        out = {"what": "pinned host frames in, pinned host results out (H2D 2.07 MB per frame), %dx%d, max_batch %d" % (W, H, max_batch)}
        call_all = lambda: lib.sift_hip_batch(hs, 1, ps, n, C.c_void_p(p_kps), C.c_void_p(p_desc), th)
        call_top = lambda: lib.sift_hip_batch_top(hs, 1, ps, n, NF, C.c_void_p(p_fk), C.c_void_p(p_fi), C.c_void_p(p_fd), C.c_void_p(p_nv), th)
        out["sift_hip_batch"] = timed(call_all)
        out["sift_hip_batch_top"] = timed(call_top)
        # the host-side results of the last sift_hip_batch call against the fixture
        if expected is not None:
            from sift_amd.checksum import words_checksum_np
            bad = [s for i, s in enumerate(seeds[:16])
                   if "%016x" % words_checksum_np(kps[i]) != expected[s - 1][1] or "%016x" % words_checksum_np(desc[i]) != expected[s - 1][2]]
            if bad:
                raise SystemExit("wrong result (sift_hip_batch, host buffers): seeds %r differ from the oracle" % bad)
            out["checked_frames"] = len(seeds[:16])
        # the sparse regime (configs[0]: the reference's test image, 218 of 10,000 slots filled): every slot and descriptor
        # row still travels back in sift_hip_batch (the contract: all 10,000 x 72 words written), only the 1000 strongest
        # in sift_hip_batch_top -- whose rate is what a valid-rows-only copy could reach at most
        if (W, H) == (1920, 1080):
            from PIL import Image
            g = os.path.join(ROOT, "tests", "golden")
            im0 = np.asarray(Image.open(os.path.join(g, "im0_gray_1920x1080.png")), dtype=np.uint8)
            for i in range(n):
                imgs[i] = im0
            sp = {"workload": "%d copies of the reference test image im0 (218 keypoints per frame)" % n,
                  "sift_hip_batch": timed(call_all), "sift_hip_batch_top": timed(call_top)}
            if th == 15:
                z = np.load(os.path.join(g, "im0_expected.npz"))
                nk = len(z["kps"])
                k0 = kps[n - 1].view(sift_amd.KP_DTYPE).reshape(K)
                d0 = desc[n - 1].reshape(K, 72)
                if not ((k0[:nk] == z["kps"]).all() and (d0[:nk] == z["desc"]).all() and not d0[nk:].any() and int(nv[n - 1]) == nk):
                    raise SystemExit("wrong result (sift_hip_batch on im0, host buffers)")
                sp["checked_frames"] = 1
            out["sparse"] = sp
        ctx.close()
        return out
    finally:
        for p in bufs:
            lib.sift_hip_host_free(p)


def main():
    argv = sys.argv[1:]
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a, argv))

    import torch
    import sift_amd
    from sift_amd.shard import DistEnv, check_ranks, device_identity, frame_range, frame_seeds, pick_device

    # SIFT_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N>1 path on a
    # one-GPU box; RCCL refuses two ranks on one device).  The driver's runs use RCCL ("nccl").
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU: there is no CPU fallback")
    backend = os.environ.get("SIFT_BENCH_BACKEND") or None
    dev = pick_device(int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count(), share=(backend == "gloo"))
    env = DistEnv(backend=backend, device=dev)
    if env.world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, env.world))
    torch.cuda.set_device(dev)
    # before anything is timed: the process group (RCCL on a node) holds the N ranks asked for, each on a GPU of its own
    identities = env.gather_ints(device_identity(torch.cuda.get_device_properties(dev), dev))
    try:
        check_ranks(a.gpus, env.dist_world_size(), identities, shared_ok=(backend == "gloo"))
    except RuntimeError as e:
        raise SystemExit("bench.py: %s" % e)
    devices = env.gather_ints(dev)
    W, H, K, th = a.width, a.height, sift_amd.MAX_KEYPOINTS, a.contrast_th

    if a.scaling == "weak":
        seeds = frame_seeds(1, a.batch, env.rank)
    else:
        lo, hi = frame_range(a.batch, env.rank, env.world)           # contiguous share of one batch
        seeds = list(range(1 + lo, 1 + hi))
        if not seeds:
            raise SystemExit("--scaling strong: --batch %d leaves rank %d without frames" % (a.batch, env.rank))
    B = len(seeds)
    ctx, step, (d_img, d_kps, d_desc, d_tot) = run_config(sift_amd, torch, dev, W, H, seeds, a.steps, a.warmup, th,
                                                            profile=not a.no_profile_events)

    env.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    env.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = env.max_over_ranks(elapsed)
    nframes = env.sum_over_ranks(B * a.steps)

    # parity of what was just timed: every frame of every rank against the oracle's fixture
    totals = d_tot.cpu().numpy()
    expected = expected_for(W, H, th)
    checked = 0 if a.no_check else check_outputs(seeds, d_kps, d_desc, totals, expected, "rank %d" % env.rank)
    checked = env.sum_over_ranks(checked)

    stage = spread = None
    if not a.no_profile_events:
        stage, spread = stage_summary(sift_amd, ctx)

    # The same K steps again with the steps alternating between two contexts (two streams, two sets
    # of scratch and output buffers): consecutive batches overlap on the GPU -- the texture-bound
    # describe kernel of one batch runs beside the VALU-bound blur of the next.  Reported as
    # "overlapped"; `value` and the per-kernel times above are the plain one-stream loop.
    overlapped = None
    if a.overlap:
        ctx2 = sift_amd.SiftContext(W, H, device=dev, max_batch=B)
        d_kps2, d_desc2, d_tot2 = torch.zeros_like(d_kps), torch.zeros_like(d_desc), torch.zeros_like(d_tot)
        ctx.set_profiling(0)
        pair = ((ctx, d_kps, d_desc, d_tot), (ctx2, d_kps2, d_desc2, d_tot2))

        def step2(k):
            c, k_, d_, t_ = pair[k & 1]
            c.detect_and_compute_dev(d_img, B, k_, d_, t_, th)

        for k in range(2):
            step2(k)
        ctx.sync(); ctx2.sync()
        env.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            step2(k)
        ctx.sync(); ctx2.sync()
        torch.cuda.synchronize()
        env.barrier()
        el2 = env.max_over_ranks(time.perf_counter() - t0)
        ok2 = bool((d_tot2.cpu().numpy() == totals).all() and torch.equal(d_kps2, d_kps) and torch.equal(d_desc2, d_desc))
        overlapped = {"streams": 2, "value": nframes / el2, "unit": "frames/s", "ms_per_step": 1e3 * el2 / a.steps,
                      "outputs_equal_one_stream": ok2}
        ctx2.close()
        del d_kps2, d_desc2, d_tot2

    # rank 0 of a one-GPU run: the CPU baseline (whose outputs are compared with the GPU's, frame by
    # frame and word by word), the other configurations and the PCIe-inclusive passes
    cpu = cpu_all = None
    oracle_checked = 0
    if a.gpus == 1 and not a.no_cpu_baseline:
        cpu, res = cpu_baseline(W, H)
        if a.scaling == "weak" and th == 15 and not a.no_check:
            for i, (rk, rd, total) in enumerate(res[:B]):
                n = int((rk["contrast"] > 0).sum())
                gk = d_kps[i].cpu().numpy().view(sift_amd.KP_DTYPE).reshape(K)
                gd = d_desc[i].cpu().numpy().view(np.uint32)
                if not ((gk == rk).all() and (gd[:n] == rd[:n]).all() and not gd[n:].any() and int(totals[i]) == total):
                    raise SystemExit("wrong result: frame of seed %d differs from the oracle run beside it" % seeds[i])
                oracle_checked += 1
        cpu_all = cpu_baseline_parallel(W, H)
    ctx.close()
    del d_kps, d_desc, d_img
    torch.cuda.empty_cache()
    others = e2e = None
    if a.gpus == 1 and (W, H) == (1920, 1080) and not a.no_other_configs:
        with_cpu = not a.no_cpu_baseline
        others = [im0_config(sift_amd, torch, dev, th, with_cpu=with_cpu),
                  other_config(sift_amd, torch, dev, 640, 480, 256, th), other_config(sift_amd, torch, dev, 3840, 2160, 64, th)]
        torch.cuda.empty_cache()
        others.append(octave_config(sift_amd, torch, dev, 3840, 2160, 64, th, with_cpu=with_cpu))
        torch.cuda.empty_cache()
    if a.gpus == 1 and not a.no_end_to_end:
        e2e = end_to_end(sift_amd, dev, W, H, th, None if a.no_check else expected)

    if env.rank == 0:
        kp_mean = float(np.minimum(totals, K).mean())
        out = {
            "metric": "1080p frames/sec (keypoints+descriptors)" if (W, H) == (1920, 1080)
                      else "%dx%d frames/sec (keypoints+descriptors)" % (W, H),
            "value": nframes / elapsed, "unit": "frames/s", "n_gpus": a.gpus, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%dx%d synthetic textured frames, full pipeline (configs[2])" % (W, H),
                       "frames_per_step_per_gpu": B, "frames_per_step": nframes // a.steps, "contrast_th": th,
                       "max_keypoints": K, "keypoints_per_frame_mean": kp_mean,
                       "parallelism": "frames x%d" % a.gpus, "resident": "inputs and outputs in HBM"},
            "ranks_seen": env.dist_world_size(), "devices": devices, "distinct_gpus": len(set(identities)),
            "backend": env.backend if env.world > 1 else None,
            "checked_frames": checked,
            "checked_what": "total, keypoint array and descriptor matrix of every frame of the last step against the oracle's fixture "
                            "(tests/golden/bench_expected.json)" + ("; %d frames also word by word against the oracle run in this process" % oracle_checked if oracle_checked else ""),
        }
        if stage is not None:
            # algorithmic bytes per launch (SURVEY.md 8(d) per-frame figures x B frames)
            alg = {
                "k_blur": 5.0 * W * H * B,                       # read u8 image, write {src,G1,G2,G3}: 5 B per pixel
                "k_detect": 6.0 * W * H * B,                     # read the 4 B/pixel planes, write 2 B/pixel gradients
                "k_describe": (min(20164.0 * kp_mean, 4.0 * W * H) + 2880000.0 + 80000.0) * B,
            }
            dom = max(alg, key=lambda k: stage[k])               # the kernel that takes the most time
            counters = load_counters("%dx%dx%d" % (W, H, B))
            traffic = counters.get("pmc_traffic.json", {})
            units = counters.get("pmc_units.json", {})

            def roof(k):
                ach = alg[k] / (stage[k] * 1e-3) / 1e9
                t = traffic.get(k)
                return {"kernel": k, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": float(t["hbm_bytes_per_launch"]) if t else None,
                        "algorithmic_bytes_per_launch": alg[k], "avg_launch_ms": stage[k]}

            def binding(k):
                """what binds kernel k: VALU issue (wave-instructions x 4 issue cycles over the SIMDs' cycles in the
                measured launch time) and the texture addressers' busy share, from the committed counter passes"""
                u = units.get(k)
                if not u:
                    return None
                clk = units.get("sustained_clock_hz", PEAK_CLOCK_HZ)
                t = stage[k] * 1e-3
                r = {"kernel": k, "valu_wave_insts_per_launch": u["SQ_INSTS_VALU"],
                     "valu_issue_frac_at_sustained_clock": u["SQ_INSTS_VALU"] * 4.0 / (NSIMD * t * clk),
                     "valu_issue_frac_at_2.4GHz": u["SQ_INSTS_VALU"] * 4.0 / (NSIMD * t * PEAK_CLOCK_HZ),
                     "sustained_clock_hz": clk, "issue_cycles_per_inst": 4}
                for name in ("ta_busy_frac", "mfma_busy_frac", "lds_active_frac"):
                    if name in u:
                        r[name] = u[name]
                if "SQ_INSTS_LDS" in u:
                    # a CU takes one ds_add_u32 / ds_write_b32 wave-instruction per 4 cycles (address + data register
                    # transfer; measured 4.0 - 4.2 with tools/ubench/lds_add_rate.hip), whatever the SIMD it comes from
                    r["lds_wave_insts_per_launch"] = u["SQ_INSTS_LDS"]
                    r["lds_issue_frac_at_sustained_clock"] = u["SQ_INSTS_LDS"] * 4.0 / (NCU * t * clk)
                shares = {"valu issue": r["valu_issue_frac_at_sustained_clock"], "matrix cores": r.get("mfma_busy_frac", 0.0),
                          "LDS instruction path": r.get("lds_issue_frac_at_sustained_clock", 0.0)}
                r["binds"] = max(shares, key=shares.get)
                if k == "k_describe":
                    # the shares are model figures (4 cycles per instruction); timing-only ablations say what they are worth
                    r["ablations"] = ("DESIGN.md 6.0: gather path alone 2.62 ms, VALU path alone 2.16 - 2.31 ms, whole kernel 2.39 ms per "
                                      "256 frames; without any ds_add 2.32 ms -- balanced, no single path binds")
                return r
            def roof_blur_mfma():
                """k_blur runs on the matrix cores: its bound is MFMA issue, not HBM.  `achieved` counts the
                ALGORITHMIC integer operations (507 MAC = 1014 op per pixel); the banded products
                issue 2304 MAC per pixel (36 MFMAs of 16x16x64 per 256 pixels and input row: an operand is two
                input rows x 32 pixels, a result tile two output rows x 8 pixels), which is what
                keeps the matrix cores busy (mfma_busy_frac, from the committed counter pass)."""
                t = stage["k_blur"] * 1e-3
                ach = 2.0 * 507.0 * W * H * B / t / 1e12
                r = {"kernel": "k_blur", "bound": "mfma", "achieved": ach, "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s (i8)",
                     "frac": ach / I8_MFMA_PEAK_TOPS, "traffic": float(traffic["k_blur"]["hbm_bytes_per_launch"]) if traffic.get("k_blur") else None,
                     "algorithmic_ops_per_launch": 2.0 * 507.0 * W * H * B, "issued_macs_per_pixel": 36 * 16384 // 256, "avg_launch_ms": stage["k_blur"],
                     "hbm": roof("k_blur")}
                if units.get("k_blur", {}).get("mfma_busy_frac") is not None:
                    r["mfma_busy_frac"] = units["k_blur"]["mfma_busy_frac"]
                return r
            out["roofline"] = roof(dom) if dom != "k_blur" else roof_blur_mfma()
            out["roofline_other"] = [roof(k) if k != "k_blur" else roof_blur_mfma() for k in alg if k != dom]
            out["roofline_binding"] = binding(dom)
            out["roofline_binding_other"] = [b for b in (binding(k) for k in alg if k != dom) if b]
            out["kernel_ms_per_step"] = stage
            out["step_kernel_ms"] = spread
        if overlapped is not None:
            out["overlapped"] = overlapped
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["cpu_baseline_all_cores"] = cpu_all
        if others is not None:
            out["other_configs"] = others
        if e2e is not None:
            out["end_to_end"] = e2e
        print(json.dumps(out), flush=True)
    env.close()


if __name__ == "__main__":
    main()
