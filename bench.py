#!/usr/bin/env python3
"""bench.py -- 1080p frames/s of the full SIFT hot path (keypoints + descriptors) on MI355X.

A step = one pass of sift_detect_and_compute (HW_SIFT.cpp:16-50) over one batch of synthetic
1920x1080 frames that is already resident in HBM; outputs stay in HBM.  One process per GPU;
for N > 1 every rank processes its own batch (frames are independent: weak scaling, no
data-path collective) and the job time is the max over ranks.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s
# v_dot4_u32_u8: 4 MAC per lane, one wave-instruction per 4 cycles per SIMD (measured with
# tools/ubench/valu_rate.hip: 16 lanes/clk, like every VOP3 op): 256 CU x 4 SIMD x 16 x 4 x 2.4 GHz
DOT4_PEAK_TMACS = 256 * 4 * 16 * 4 * 2.4e9 / 1e12
EXPECTED_KP_1080P_SEED1 = 9493              # tests/golden/synth_expected.json, SURVEY.md 8(d) cfg 3


def cpu_baseline(width, height, budget_s=12.0, max_frames=12):
    """The CPU oracle (a port of the reference's algorithm, oracle/sift_oracle.c) on one host core,
    on a bounded sample of the same workload.  This is the only place bench.py touches oracle/."""
    from oracle import oracle as orc
    orc.build()
    frames = [orc.synth_frame(1 + i, width, height) for i in range(max_frames)]
    orc.detect_and_compute(frames[0])       # page in
    n, t0 = 0, time.perf_counter()
    while n < max_frames and (n < 3 or time.perf_counter() - t0 < budget_s):
        orc.detect_and_compute(frames[n])
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d synthetic %dx%d frames (seeds 1..%d), full pipeline, %.1f s" % (n, width, height, n, dt)}


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


def load_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json,
    written by tools/pmc_traffic.py from separate rocprofv3 --pmc runs), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            d = json.load(f)
        e = d.get(workload, {}).get(kernel)
        return float(e["hbm_bytes_per_launch"]) if e else None
    except (OSError, ValueError, KeyError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per step per GPU (the kernels reach their steady rate from ~200 frames on)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--contrast-th", type=int, default=15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the keypoint-count sanity check (kernel experiments)")
    ap.add_argument("--no-overlap", action="store_true", help="skip the two-stream pass")
    ap.add_argument("--no-profile-events", action="store_true", help="leave the per-kernel HIP events out")
    a = ap.parse_args()

    import torch
    import sift_amd
    from sift_amd.shard import DistEnv, frame_seeds

    # SIFT_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N>1 path on a
    # one-GPU box; RCCL refuses two ranks on one device).  The driver's runs use RCCL ("nccl").
    env = DistEnv(backend=os.environ.get("SIFT_BENCH_BACKEND") or None)
    if env.world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (a.gpus, env.world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU: there is no CPU fallback")
    dev = env.local_rank
    if dev >= torch.cuda.device_count():
        if env.backend != "gloo":
            raise SystemExit("rank %d has no GPU (%d visible)" % (env.rank, torch.cuda.device_count()))
        dev %= torch.cuda.device_count()
    torch.cuda.set_device(dev)
    W, H, B, K = a.width, a.height, a.batch, sift_amd.MAX_KEYPOINTS

    seeds = frame_seeds(1, B, env.rank)
    frames = np.stack([sift_amd.synth_frame(s, W, H) for s in seeds])
    d_img = torch.from_numpy(frames).to("cuda:%d" % dev)
    d_kps = torch.zeros((B, K, 2), dtype=torch.int32, device=d_img.device)
    d_desc = torch.zeros((B, K, 72), dtype=torch.int32, device=d_img.device)
    d_tot = torch.zeros(B, dtype=torch.int32, device=d_img.device)
    ctx = sift_amd.SiftContext(W, H, device=dev, max_batch=B)

    def step():
        ctx.detect_and_compute_dev(d_img, B, d_kps, d_desc, d_tot, a.contrast_th)

    # spin-up, before the W warm-up steps: the GPU's clocks take tens of milliseconds of load to
    # settle (the first steps after an idle period run several per cent slower)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.25:
        step()
        ctx.sync()
    for _ in range(a.warmup):
        step()
    ctx.sync()
    if not a.no_profile_events:
        ctx.set_profiling(max(1, min(a.steps, 4096)))

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

    # sanity (parity proper lives in tests/): the seed-1 frame must give the oracle's keypoint count
    totals = d_tot.cpu().numpy()
    if (env.rank == 0 and not a.no_check and (W, H, a.contrast_th) == (1920, 1080, 15)
            and int(totals[0]) != EXPECTED_KP_1080P_SEED1):
        raise SystemExit("wrong result: %d keypoints on the seed-1 frame, expected %d"
                         % (int(totals[0]), EXPECTED_KP_1080P_SEED1))

    stage = None
    if not a.no_profile_events:
        recs = ctx.stage_ms_all()
        stage = {k: float(np.mean([r[k] for r in recs])) for k in sift_amd.STAGE_NAMES}
        # spread over the timed steps: sum of the five kernels of a step (HIP events), per frame too
        per_step = np.array([sum(r[k] for k in sift_amd.STAGE_NAMES) for r in recs])
        spread = {"p5": float(np.percentile(per_step, 5)), "p50": float(np.percentile(per_step, 50)),
                  "p95": float(np.percentile(per_step, 95)), "steps": int(len(per_step)),
                  "what": "sum of the five kernels' HIP-event times per step, ms"}

    # The same K steps again with the steps alternating between two contexts (two streams, two sets
    # of scratch and output buffers): consecutive batches overlap on the GPU -- the texture-bound
    # describe kernel of one batch runs beside the VALU-bound blur of the next.  Reported as
    # "overlapped"; `value` and the per-kernel times above are the plain one-stream loop.
    overlapped = None
    if not a.no_overlap:
        ctx2 = sift_amd.SiftContext(W, H, device=dev, max_batch=B)
        d_kps2, d_desc2, d_tot2 = torch.zeros_like(d_kps), torch.zeros_like(d_desc), torch.zeros_like(d_tot)
        ctx.set_profiling(0)
        pair = ((ctx, d_kps, d_desc, d_tot), (ctx2, d_kps2, d_desc2, d_tot2))

        def step2(k):
            c, k_, d_, t_ = pair[k & 1]
            c.detect_and_compute_dev(d_img, B, k_, d_, t_, a.contrast_th)

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

    ctx.close()

    if env.rank == 0:
        out = {
            "metric": "1080p frames/sec (keypoints+descriptors)" if (W, H) == (1920, 1080)
                      else "%dx%d frames/sec (keypoints+descriptors)" % (W, H),
            "value": nframes / elapsed, "unit": "frames/s", "n_gpus": a.gpus, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%dx%d synthetic textured frames, full pipeline (configs[2])" % (W, H),
                       "frames_per_step_per_gpu": B, "contrast_th": a.contrast_th, "max_keypoints": K,
                       "keypoints_per_frame_mean": float(np.minimum(totals, K).mean()),
                       "parallelism": "frames x%d" % a.gpus, "resident": "inputs and outputs in HBM"},
        }
        if stage is not None:
            kp_mean = float(np.minimum(totals, K).mean())
            # algorithmic bytes per launch (SURVEY.md 8(d) per-frame figures x B frames)
            alg = {
                "k_blur": 5.0 * W * H * B,                       # read u8 image, write {src,G1,G2,G3}: 5 B per pixel
                "k_detect": 6.0 * W * H * B,                     # read the 4 B/pixel planes, write 2 B/pixel gradients
                "k_describe": (min(20164.0 * kp_mean, 4.0 * W * H) + 2880000.0 + 80000.0) * B,
            }
            dom = max(alg, key=lambda k: stage[k])               # the kernel that takes the most time
            wl = "%dx%dx%d" % (W, H, B)

            def roof(k):
                ach = alg[k] / (stage[k] * 1e-3) / 1e9
                return {"kernel": k, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": load_traffic(k, wl),
                        "algorithmic_bytes_per_launch": alg[k], "avg_launch_ms": stage[k]}
            out["roofline"] = roof(dom)
            out["roofline_other"] = [roof(k) for k in alg if k != dom]
            t_blur = stage["k_blur"] * 1e-3
            macs = 507.0 * W * H * B
            out["valu"] = {"kernel": "k_blur", "achieved": macs / t_blur / 1e12, "peak": DOT4_PEAK_TMACS,
                           "unit": "TMAC/s (u8 dot4)", "frac": macs / t_blur / 1e12 / DOT4_PEAK_TMACS}
            out["kernel_ms_per_step"] = stage
            out["step_kernel_ms"] = spread
        if overlapped is not None:
            out["overlapped"] = overlapped
        if a.gpus == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H)
            out["cpu_baseline_all_cores"] = cpu_baseline_parallel(W, H)
        print(json.dumps(out), flush=True)
    env.close()


if __name__ == "__main__":
    main()
