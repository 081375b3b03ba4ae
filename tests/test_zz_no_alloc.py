"""The entry point that allocates nothing (sift_hip_detect_and_compute_static) against the existing
device-buffer entry point (sift_hip_detect_and_compute_dev), through the C ABI, bit for bit.

Two calls in a row on the same GPU, the second with a smaller input: the static array keeps the
first call's contents, so the second call shows that what must start at zero is cleared.  Collected
last (file name) so that a failure here cannot hide another test behind `pytest -x`.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = 10000


def _static_call(sift, torch, imgs, th=15, stream=None):
    lib = sift.load_library()
    B, H, W = imgs.shape
    d_img = torch.from_numpy(imgs).cuda()
    d_kps = torch.full((B, K, 2), -1, dtype=torch.int32, device="cuda")
    d_desc = torch.full((B, K, 72), -1, dtype=torch.int32, device="cuda")
    d_tot = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = lib.sift_hip_detect_and_compute_static(0, C.c_void_p(d_img.data_ptr()), B, W, H, C.c_void_p(d_kps.data_ptr()),
                                                C.c_void_p(d_desc.data_ptr()), C.c_void_p(d_tot.data_ptr()), th,
                                                C.c_void_p(stream) if stream else None)
    sift.capi.check(rc)
    torch.cuda.synchronize()
    return d_kps, d_desc, d_tot


def _context_call(sift, torch, imgs, th=15):
    B, H, W = imgs.shape
    d_img = torch.from_numpy(imgs).cuda()
    d_kps = torch.full((B, K, 2), -1, dtype=torch.int32, device="cuda")
    d_desc = torch.full((B, K, 72), -1, dtype=torch.int32, device="cuda")
    d_tot = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    with sift.SiftContext(W, H, max_batch=B) as ctx:
        ctx.detect_and_compute_dev(d_img, B, d_kps, d_desc, d_tot, th)
        ctx.sync()
    return d_kps, d_desc, d_tot


def test_static_entry_point_matches_the_context_entry_point(sift, orc):
    import torch
    lib = sift.load_library()
    big = np.stack([orc.synth_frame(5, 1920, 1080)])
    small = np.stack([orc.synth_frame(60 + s, 640, 480) for s in range(3)])
    tiny = np.stack([orc.synth_frame(9, 324, 201)])
    assert 0 < lib.sift_hip_static_scratch_needed(1920, 1080, 1) <= (32 << 20)
    assert lib.sift_hip_static_scratch_needed(1920, 1080, 2) > (32 << 20)
    assert lib.sift_hip_static_scratch_needed(1921, 1080, 1) == 0
    st = torch.cuda.Stream()
    # two calls in a row, the second smaller (and with more frames), then smaller again on a caller's stream
    for imgs, stream in ((big, None), (small, None), (tiny, st.cuda_stream), (big, None)):
        got = _static_call(sift, torch, imgs, stream=stream)
        want = _context_call(sift, torch, imgs)
        for g, w, what in zip(got, want, ("keypoints", "descriptors", "totals")):
            assert torch.equal(g, w), "%s differ at %r" % (what, imgs.shape)
        assert int(got[2].min().item()) > 0                         # every frame has keypoints: the comparison is not of zeros
    # the small frames also against the oracle (the checker of every other GPU test)
    d_kps, d_desc, d_tot = _static_call(sift, torch, small)
    kps = d_kps.cpu().numpy().view(sift.KP_DTYPE).reshape(len(small), K)
    desc = d_desc.cpu().numpy().view(np.uint32)
    for b in range(len(small)):
        rk, rd, total = orc.detect_and_compute(small[b])
        n = int((rk["contrast"] > 0).sum())
        assert (kps[b] == rk).all() and (desc[b, :n] == rd[:n]).all() and not desc[b, n:].any()
        assert int(d_tot[b].item()) == total


def test_static_entry_point_alternating_sizes_under_capture(sift, orc):
    """The tables of the two most recently used widths stay in the array: a caller that alternates two frame sizes
    rebuilds nothing, so those calls can be recorded into a HIP graph (a call that had to allocate, copy tables from the
    host or synchronise would fail the capture) -- and the library's own creation counters stay where they were."""
    import torch
    from sift_amd.hw_sift import created_counts
    a = np.stack([orc.synth_frame(31, 640, 480)])
    b = np.stack([orc.synth_frame(32, 324, 201)])
    want = {id(x): _context_call(sift, torch, x) for x in (a, b)}
    for imgs in (a, b):                                                # both widths' tables become resident
        _static_call(sift, torch, imgs)
    before = created_counts()
    lib = sift.load_library()
    st = torch.cuda.Stream()
    for imgs in (a, b, a, b):
        B, H, W = imgs.shape
        d_img = torch.from_numpy(imgs).cuda()
        d_kps = torch.full((B, K, 2), -1, dtype=torch.int32, device="cuda")
        d_desc = torch.full((B, K, 72), -1, dtype=torch.int32, device="cuda")
        d_tot = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            sift.capi.check(lib.sift_hip_detect_and_compute_static(0, C.c_void_p(d_img.data_ptr()), B, W, H, C.c_void_p(d_kps.data_ptr()),
                                                                   C.c_void_p(d_desc.data_ptr()), C.c_void_p(d_tot.data_ptr()), 15,
                                                                   C.c_void_p(st.cuda_stream)))
        g.replay()
        torch.cuda.synchronize()
        for got, w, what in zip((d_kps, d_desc, d_tot), want[id(imgs)], ("keypoints", "descriptors", "totals")):
            assert torch.equal(got, w), "%s differ at %r" % (what, imgs.shape)
    assert created_counts() == before


def test_static_entry_point_rejects_what_does_not_fit(sift, orc):
    import torch
    two = np.stack([orc.synth_frame(1, 1920, 1080)] * 2)
    with pytest.raises(sift.SiftHipError) as e:
        _static_call(sift, torch, two)
    assert e.value.code == -1 and "static array" in str(e.value)
    lib = sift.load_library()
    assert lib.sift_hip_detect_and_compute_static(0, None, 1, 640, 480, None, None, None, 15, None) == -1
