"""Further octaves ("an optional multi-octave mode", an extension).  The reference processes one
octave and holds nothing that a second one could be pinned to, so the semantics are the ones include/sift_hip.h
states: the next octave's input is pixel (2x, 2y) of a blurred plane of the frame just processed, and it goes through
the unchanged single-octave path.  The oracle restates that composition; parity unpinned by construction."""
import numpy as np
import pytest


def test_oracle_next_octave_is_the_decimated_blur(orc):
    img = orc.synth_frame(3, 151, 147)
    g = orc.blur(img)
    for plane in (1, 2, 3):
        o = orc.next_octave(img, plane)
        assert o.shape == (73, 75) and o.dtype == np.uint8
        for y2, x2 in ((0, 0), (72, 74), (10, 33), (71, 1)):
            assert o[y2, x2] == g[plane - 1][2 * y2, 2 * x2]
    outs = orc.detect_and_compute_octaves(orc.synth_frame(1, 640, 480), 3)
    assert [k.shape for k, _, _ in outs] == [(10000,)] * 3
    assert outs[0][2] == 985                                        # octave 0 is the reference's pipeline
    assert 0 < outs[1][2] < outs[0][2]


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(640, 480), (1920, 1080), (643, 481), (150, 149), (2, 2)])
def test_next_octave_frames_against_the_oracle(sift, orc, W, H):
    img = orc.synth_frame(5, W, H)
    with sift.SiftContext(W, H) as ctx:
        ctx.sift_detect_and_compute(img, 15)
        for plane in (1, 2, 3):
            got = ctx.next_octave(0, plane)
            assert got.shape == (H // 2, W // 2)
            assert (got == orc.next_octave(img, plane)).all(), plane
        with pytest.raises(sift.SiftHipError):
            ctx.next_octave(0, 4)
        with pytest.raises(sift.SiftHipError):
            ctx.next_octave(1, 2)                                   # the last call had one frame


@pytest.mark.gpu
def test_one_pixel_frames_have_no_next_octave(sift, orc):
    with sift.SiftContext(1, 1) as ctx:
        ctx.sift_detect_and_compute(np.zeros((1, 1), np.uint8), 15)
        with pytest.raises(sift.SiftHipError):
            ctx.next_octave(0, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,noct", [(1920, 1080, 3), (1283, 722, 4)])
def test_octave_chain_against_the_oracle(sift, orc, W, H, noct):
    img = orc.synth_frame(2, W, H)
    got = sift.sift_detect_and_compute_octaves(img, noct, 15)
    want = orc.detect_and_compute_octaves(img, noct, 15)
    assert len(got) == noct
    totals = []
    for (kps, desc), (rk, rd, total) in zip(got, want):
        n = int((rk["contrast"] > 0).sum())
        assert (kps == rk).all()
        assert (desc[:n] == rd[:n]).all() and not desc[n:].any()
        totals.append(n)
    assert totals[0] > totals[1] > 0                                # the half-size octave does find keypoints


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,B", [(640, 480, 3), (3840, 2160, 2)])
def test_batched_octave_chain_on_the_device(sift, orc, W, H, B):
    """a few frames, two octaves, device buffers only: ctx0 -> next_octave_dev -> ctx1; the second case is
    BASELINE.json configs[3] ("4K frames, extra octave"), the shape bench.py times"""
    import torch
    frames = np.stack([orc.synth_frame(20 + i, W, H) for i in range(B)])
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(frames).to(dev)
    with sift.SiftContext(W, H, max_batch=B) as c0, sift.SiftContext(W // 2, H // 2, max_batch=B) as c1:
        K = c0.max_keypoints
        d_k0 = torch.zeros((B, K, 2), dtype=torch.int32, device=dev)
        d_d0 = torch.zeros((B, K, 72), dtype=torch.int32, device=dev)
        d_k1 = torch.zeros_like(d_k0)
        d_d1 = torch.zeros_like(d_d0)
        d_img1 = torch.zeros((B, H // 2, W // 2), dtype=torch.uint8, device=dev)
        c0.detect_and_compute_dev(d_img.data_ptr(), B, d_k0.data_ptr(), d_d0.data_ptr())
        c0.next_octave_dev(0, B, d_img1.data_ptr(), plane=2)
        c0.sync()
        c1.detect_and_compute_dev(d_img1.data_ptr(), B, d_k1.data_ptr(), d_d1.data_ptr())
        c1.sync()
        for i in range(B):
            want = orc.detect_and_compute_octaves(frames[i], 2, 15)
            assert (d_img1[i].cpu().numpy() == orc.next_octave(frames[i], 2)).all()
            for d_k, d_d, (rk, rd, _) in ((d_k0, d_d0, want[0]), (d_k1, d_d1, want[1])):
                n = int((rk["contrast"] > 0).sum())
                assert (d_k[i].cpu().numpy().view(np.uint32).reshape(-1) == rk.view(np.uint32).reshape(-1)).all()
                assert (d_d[i, :n].cpu().numpy().view(np.uint32) == rd[:n]).all()
