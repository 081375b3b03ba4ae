"""Post-processing of main.cpp:82-130 (next row (f)1 of the scope table): the MAX_FINAL_KEYPOINTS
keypoints of largest contrast in descending order.

The reference uses std::sort, which is not stable: among equal contrasts (contrast is |DoG2| <= 255,
so ties are the rule) its order is whatever the library's introsort leaves.  What every correct
implementation agrees on is checked as such (the multiset of contrasts per rank, the strict part of
the order); the order inside ties is pinned to slot order by the oracle and checked bit-exactly."""
import numpy as np
import pytest


def make_kps(orc, rng, nvalid, n=10000, cmax=256, cmin=15):
    k = np.zeros(n, orc.KP_DTYPE)
    k["x"][:nvalid] = rng.integers(72, 1848, nvalid)
    k["y"][:nvalid] = rng.integers(72, 1008, nvalid)
    k["contrast"][:nvalid] = rng.integers(cmin, cmax, nvalid, dtype=np.uint64).astype(np.uint32)
    return k


def numpy_select(k, nfinal):
    order = np.argsort(-k["contrast"].astype(np.int64), kind="stable")[:nfinal]
    return k[order], order.astype(np.uint32), int(((k[order]["x"] > 0) & (k[order]["y"] > 0)).sum())


CASES = [(9493, 1000, 256), (985, 1000, 256), (0, 1000, 256), (10000, 1000, 16), (10000, 1, 256), (10000, 4096, 256),
         (5000, 1024, 2 ** 32), (3000, 1025, 2 ** 20), (10000, 1000, 2 ** 32)]


@pytest.mark.parametrize("nvalid,nfinal,cmax", CASES)
def test_oracle_select_equals_stable_sort(orc, nvalid, nfinal, cmax):
    rng = np.random.default_rng(nvalid + nfinal)
    k = make_kps(orc, rng, nvalid, cmax=cmax)
    f, i, nv = orc.select_top(k, nfinal)
    rf, ri, rnv = numpy_select(k, nfinal)
    assert (f == rf).all() and (i == ri).all() and nv == rnv
    # what std::sort guarantees whatever it does with ties: contrasts by rank, and a valid permutation
    assert (f["contrast"] == np.sort(k["contrast"])[::-1][:nfinal]).all()
    assert len(np.unique(i)) == nfinal and (k[i] == f).all()


def test_oracle_select_on_detector_output(orc):
    img = orc.synth_frame(1, 640, 480)
    kps, _, _ = orc.detect_and_compute(img)
    f, i, nv = orc.select_top(kps, 1000)
    assert nv == 985                       # tests/golden/synth_expected.json; fewer than 1000: the tail is padding
    assert not f[nv:]["contrast"].any() and (f[:nv]["contrast"] > 0).all()
    assert (np.diff(f["contrast"].astype(np.int64)) <= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("nvalid,nfinal,cmax", CASES)
def test_gpu_select_equals_oracle(sift, orc, nvalid, nfinal, cmax):
    rng = np.random.default_rng(nvalid * 7 + nfinal)
    k = make_kps(orc, rng, nvalid, cmax=cmax)
    with sift.SiftContext(64, 64) as ctx:
        f, i, nv = ctx.select_top(k, nfinal)
    rf, ri, rnv = orc.select_top(k, nfinal)
    assert (f == rf).all() and (i == ri).all() and nv == rnv


@pytest.mark.gpu
def test_gpu_select_degenerate_inputs(sift, orc):
    with sift.SiftContext(64, 64) as ctx:
        k = np.zeros(10000, orc.KP_DTYPE)
        k["x"] = 100; k["y"] = 200; k["contrast"] = 77                    # every contrast equal: slot order decides
        f, i, nv = ctx.select_top(k, 1000)
        assert (i == np.arange(1000)).all() and nv == 1000 and (f["contrast"] == 77).all()
        k["contrast"] = np.arange(10000, dtype=np.uint32) * 429496 + 5    # strictly increasing 32-bit values
        f, i, nv = ctx.select_top(k, 1000)
        assert (i == np.arange(9999, 8999, -1)).all()
        k["contrast"][1234] = 0xFFFFFFFF                                  # the largest value of the type
        f, i, nv = ctx.select_top(k, 7)
        assert i[0] == 1234 and f["contrast"][0] == 0xFFFFFFFF
        for bad in (0, 4097, 10001):
            with pytest.raises(sift.SiftHipError):
                ctx.select_top(k, bad)
    with sift.SiftContext(64, 64, max_keypoints=500) as ctx:              # cap below MAX_FINAL_KEYPOINTS
        k = make_kps(orc, np.random.default_rng(3), 300, n=500)
        f, i, nv = ctx.select_top(k, 500)
        rf, ri, rnv = orc.select_top(k, 500)
        assert (f == rf).all() and (i == ri).all() and nv == rnv == 300
        with pytest.raises(sift.SiftHipError):
            ctx.select_top(k, 501)


@pytest.mark.gpu
def test_gpu_pipeline_then_select_on_device(sift, orc):
    """detect_and_compute_dev -> select_top_dev on a batch, everything resident; descriptor rows gathered."""
    import torch
    W, H, B, K, NF = 640, 480, 3, 10000, 1000
    imgs = np.stack([orc.synth_frame(s, W, H) for s in (1, 2, 3)])
    with sift.SiftContext(W, H, max_batch=B) as ctx:
        d_img = torch.from_numpy(imgs).cuda()
        d_kps = torch.zeros((B, K, 2), dtype=torch.int32, device="cuda")
        d_desc = torch.zeros((B, K, 72), dtype=torch.int32, device="cuda")
        d_fin = torch.full((B, NF, 2), -1, dtype=torch.int32, device="cuda")
        d_idx = torch.full((B, NF), -1, dtype=torch.int32, device="cuda")
        d_nv = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        d_fdesc = torch.full((B, NF, 72), -1, dtype=torch.int32, device="cuda")
        ctx.detect_and_compute_dev(d_img, B, d_kps, d_desc)
        ctx.select_top_dev(d_kps, B, NF, d_fin, d_idx, d_nv, d_desc, d_fdesc)
        ctx.sync()
        fin = d_fin.cpu().numpy().view(sift.KP_DTYPE).reshape(B, NF)
        idx = d_idx.cpu().numpy().view(np.uint32)
        fdesc = d_fdesc.cpu().numpy().view(np.uint32)
        for b in range(B):
            rk, rd, _ = orc.detect_and_compute(imgs[b])
            rf, ri, rnv = orc.select_top(rk, NF)
            assert (fin[b] == rf).all() and (idx[b] == ri).all() and int(d_nv[b].item()) == rnv
            assert (fdesc[b, :rnv] == rd[ri[:rnv]]).all() and not fdesc[b, rnv:].any()
        with pytest.raises(sift.SiftHipError):
            ctx.select_top_dev(d_kps, B, NF, d_fin, d_idx, d_nv, d_desc, None)


@pytest.mark.gpu
def test_batch_top_is_batch_then_select(sift, orc):
    """sift_hip_batch_top: ingest -> compute -> select -> compact egress, chunks double buffered."""
    from sift_amd.hw_sift import batch_top
    W, H, NF = 320, 240, 50
    imgs = [orc.synth_frame(s, W, H) for s in range(40, 47)]
    with sift.SiftContext(W, H, max_batch=4) as c0, sift.SiftContext(W, H, max_batch=1) as c1:
        for ctxs in ([c0], [c0, c1]):
            fin, idx, fdesc, nv = batch_top(ctxs, imgs, NF)
            for f, img in enumerate(imgs):
                rk, rd, _ = orc.detect_and_compute(img)
                rf, ri, rnv = orc.select_top(rk, NF)
                assert (fin[f] == rf).all() and (idx[f] == ri).all() and int(nv[f]) == rnv
                assert (fdesc[f, :rnv] == rd[ri[:rnv]]).all() and not fdesc[f, rnv:].any()
        with pytest.raises(sift.SiftHipError):
            batch_top([c0], imgs, 0)
