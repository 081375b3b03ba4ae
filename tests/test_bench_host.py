"""Host logic of bench.py that needs no GPU: the self-launch of N ranks, the checksum both sides of
the post-run check use, the device-side frame generator and the oracle's fixture."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_launch_builds_the_torchrun_command(monkeypatch):
    bench = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    argv = ["--gpus", "4", "--steps", "5", "--scaling", "strong"]
    rc = bench.self_launch(bench.parse_args(argv), argv)
    cmd = seen["cmd"]
    assert rc == 7                                               # the children's exit code is the launcher's
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_two_ranks_without_a_gpu_fail_loudly():
    """`python bench.py --gpus 2` as a plain command starts two ranks; without a GPU each of them says so
    and the launcher's exit code is non-zero (there is no CPU fallback)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=170, cwd=ROOT,
                       env=dict(os.environ, SIFT_BENCH_BACKEND="gloo", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert r.returncode != 0
    assert "needs a gfx950 GPU" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_pick_device():
    from sift_amd.shard import pick_device
    assert [pick_device(r, 8) for r in range(8)] == list(range(8))
    assert pick_device(5, 1) == 0                                # per-rank visibility mask: one device, index 0
    assert pick_device(3, 2, share=True) == 1
    with pytest.raises(RuntimeError):
        pick_device(3, 2)
    with pytest.raises(RuntimeError):
        pick_device(0, 0)


def test_checksum_numpy_and_torch_agree():
    import torch
    from sift_amd.checksum import words_checksum_np, words_checksum_torch
    rng = np.random.default_rng(0)
    a = rng.integers(0, 2 ** 32, (5, 20000), dtype=np.uint64).astype(np.uint32)
    a[1] = 0
    a[2] = 0xFFFFFFFF
    t = torch.from_numpy(a.view(np.int32)).reshape(5, 10000, 2)
    assert words_checksum_torch(t, frames_per_chunk=2) == [words_checksum_np(a[i]) for i in range(5)]
    b = a[0].copy()
    b[[3, 4]] = b[[4, 3]]                                        # order-sensitive
    assert words_checksum_np(b) != words_checksum_np(a[0])


def test_device_frame_generator_matches_the_host_one(orc):
    from sift_amd.synth import synth_frame, synth_frames_torch
    seeds = [1, 2, 300, 2 ** 31 + 5]
    for W, H in ((640, 480), (324, 201), (36, 33)):
        t = synth_frames_torch(seeds, W, H).numpy()
        for i, s in enumerate(seeds):
            assert (t[i] == synth_frame(s, W, H)).all() and (t[i] == orc.synth_frame(s, W, H)).all()


def test_bench_fixture_is_the_oracles(orc, golden_dir):
    """tests/golden/bench_expected.json (made by make_bench_fixture.py) against the oracle run here,
    on a sample of seeds of each configuration"""
    from sift_amd.checksum import words_checksum_np
    with open(os.path.join(golden_dir, "bench_expected.json")) as f:
        exp = json.load(f)
    assert len(exp["1920x1080"]) == 2048 and len(exp["640x480"]) == 256 and len(exp["3840x2160"]) == 64
    assert exp["1920x1080"][0][0] == 9493 and exp["640x480"][0][0] == 985
    for name, seeds in (("640x480", (1, 2, 77, 256)), ("1920x1080", (1, 1500))):
        W, H = map(int, name.split("x"))
        for s in seeds:
            kps, desc, total = orc.detect_and_compute(orc.synth_frame(s, W, H), 15)
            n = int((kps["contrast"] > 0).sum())
            desc[n:] = 0
            assert exp[name][s - 1] == [total, "%016x" % words_checksum_np(kps.view(np.uint32)), "%016x" % words_checksum_np(desc)]
