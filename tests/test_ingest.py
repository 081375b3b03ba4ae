"""Ingest of main.cpp:51-58 for colour frames (next row (f)2): 24-bit BGR rows -> gray, the weights of
cv::imread(..., CV_LOAD_IMAGE_GRAYSCALE).  OpenCV is not available here: the formula is restated from
its published source (parity unpinned); what is checked is oracle == numpy restatement == GPU."""
import numpy as np
import pytest


def numpy_gray(bgr3):
    b, g, r = [bgr3[..., i].astype(np.int64) for i in range(3)]
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def test_oracle_bgr2gray(orc):
    rng = np.random.default_rng(1)
    W, H = 36, 9
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    assert (orc.bgr2gray(img, W) == numpy_gray(img)).all()
    # padded rows (BMP pads to 4 bytes) and bottom-up storage
    pitch = 3 * W + 4
    buf = rng.integers(0, 256, (H, pitch), dtype=np.uint8)
    buf[:, :3 * W] = img.reshape(H, -1)
    assert (orc.bgr2gray(buf, W) == numpy_gray(img)).all()
    assert (orc.bgr2gray(buf[::-1].copy(), W, bottom_up=True) == numpy_gray(img)).all()
    # weights sum to 2^14: gray levels are fixed points; primaries as OpenCV gives them
    for v in (0, 1, 127, 255):
        assert orc.bgr2gray(np.full((1, 4, 3), v, np.uint8), 4)[0, 0] == v
    prim = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0]]], np.uint8)   # B, G, R, B+G
    assert orc.bgr2gray(prim, 4).tolist() == [[29, 150, 76, 179]]


@pytest.mark.gpu
def test_gpu_bgr2gray_equals_oracle(sift, orc):
    rng = np.random.default_rng(2)
    for W, H, pad in ((64, 48, 0), (68, 5, 8), (1920, 1080, 0), (324, 7, 4), (321, 9, 1), (5, 3, 1), (1601, 4, 5), (642, 3, 2)):
        pitch = 3 * W + pad
        buf = rng.integers(0, 256, (H, pitch), dtype=np.uint8)
        with sift.SiftContext(W, H) as ctx:
            for bottom_up in (False, True):
                assert (ctx.gray_from_bgr(buf, bottom_up) == orc.bgr2gray(buf, W, bottom_up)).all()
            with pytest.raises(sift.SiftHipError):
                ctx.gray_from_bgr(buf[:, :3 * W - 4].copy())


@pytest.mark.gpu
def test_gpu_ingest_feeds_the_pipeline_on_device(sift, orc):
    """BGR frames resident in HBM -> gray -> keypoints + descriptors without leaving the device."""
    import torch
    W, H, B, K = 640, 480, 2, 10000
    gray = np.stack([orc.synth_frame(s, W, H) for s in (1, 2)])
    bgr = np.repeat(gray[..., None], 3, axis=3)                    # B = G = R: gray is a fixed point of the weights
    bgr[..., 0] ^= 0                                               # (keeps the frames the detector tests know)
    with sift.SiftContext(W, H, max_batch=B) as ctx:
        d_bgr = torch.from_numpy(bgr.reshape(B, H, 3 * W)).cuda()
        d_gray = torch.zeros((B, H, W), dtype=torch.uint8, device="cuda")
        d_kps = torch.zeros((B, K, 2), dtype=torch.int32, device="cuda")
        d_desc = torch.zeros((B, K, 72), dtype=torch.int32, device="cuda")
        ctx.gray_from_bgr_dev(d_bgr, B, 3 * W, False, d_gray)
        ctx.detect_and_compute_dev(d_gray, B, d_kps, d_desc)
        ctx.sync()
        assert (d_gray.cpu().numpy() == gray).all()
        kps = d_kps.cpu().numpy().view(sift.KP_DTYPE).reshape(B, K)
        for b in range(B):
            rk, _, _ = orc.detect_and_compute(gray[b])
            assert (kps[b] == rk).all()
