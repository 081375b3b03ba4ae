"""Matcher for DMatch_t (SIFTParameters.hpp:122-128).  The reference declares the type and
nothing else, so there is no reference behaviour to pin: the semantics are the ones include/sift_hip.h
states (nearest train row in the L1 sense, ties to the smaller index), restated in the oracle."""
import numpy as np
import pytest


def np_match(q, t):
    d = np.abs(q[:, None, :].astype(np.int64) - t[None, :, :].astype(np.int64)).sum(axis=2)
    return d.min(axis=1), d.argmin(axis=1)


def test_oracle_match_equals_numpy(orc):
    rng = np.random.default_rng(1)
    q = rng.integers(0, 200000, (37, 72)).astype(np.uint32)
    t = rng.integers(0, 200000, (53, 72)).astype(np.uint32)
    t[7] = t[3]                                                   # a tie: the smaller index wins
    q[5] = t[7]
    m = orc.match(q, t)
    dist, idx = np_match(q, t)
    assert (m["distance"] == dist).all() and (m["trainIdx"] == idx).all() and (m["queryIdx"] == np.arange(37)).all()
    assert m["trainIdx"][5] == 3 and m["distance"][5] == 0
    e = orc.match(q, np.zeros((0, 72), np.uint32))
    assert (e["trainIdx"] == -1).all() and (e["distance"] == 0xFFFFFFFF).all()


@pytest.mark.gpu
def test_match_against_the_oracle(sift, orc):
    import torch
    rng = np.random.default_rng(2)
    with sift.SiftContext(640, 480) as ctx:
        for nq, nt in ((1, 1), (63, 5), (64, 64), (65, 3), (1000, 1000), (300, 0), (0, 10), (2500, 777)):
            q = rng.integers(0, 190000, (nq, 72)).astype(np.uint32)
            t = rng.integers(0, 190000, (nt, 72)).astype(np.uint32)
            if nt > 8 and nq > 8:
                t[7] = t[3]
                q[5] = t[7]
                q[6, 0] = 4000000000                               # a distance near the top of 32 bits (sums stay below 2^32)
            got = ctx.match(q, t)
            want = orc.match(q, t)
            assert (got == want).all(), (nq, nt)
        # real descriptors: a frame against itself moved by (5, 3) -- most keypoints find their moved twin
        a = orc.synth_frame(9, 640, 480)
        b = np.zeros_like(a)
        b[3:, 5:] = a[:-3, :-5]
        ka, da = ctx.sift_detect_and_compute(a)
        kb, db = ctx.sift_detect_and_compute(b)
        na, nb = int((ka["contrast"] > 0).sum()), int((kb["contrast"] > 0).sum())
        m = ctx.match(da[:na], db[:nb])
        assert (m == orc.match(da[:na], db[:nb])).all()
        hit = (ka["x"][:na] + 5 == kb["x"][m["trainIdx"]]) & (ka["y"][:na] + 3 == kb["y"][m["trainIdx"]])
        assert hit.mean() > 0.9 and (m["distance"][hit] == 0).all()
        # several pairs per launch, device buffers
        P, nq, nt = 5, 130, 90
        q = rng.integers(0, 100000, (P, nq, 72)).astype(np.uint32)
        t = rng.integers(0, 100000, (P, nt, 72)).astype(np.uint32)
        d_q, d_t = torch.from_numpy(q.view(np.int32)).cuda(), torch.from_numpy(t.view(np.int32)).cuda()
        d_m = torch.zeros((P, nq, 2), dtype=torch.int32, device="cuda")
        ctx.match_dev(d_q, nq, d_t, nt, P, d_m)
        ctx.sync()
        got = d_m.cpu().numpy().view(sift.DMATCH_DTYPE).reshape(P, nq)
        for p in range(P):
            assert (got[p] == orc.match(q[p], t[p])).all(), p
        with pytest.raises(sift.SiftHipError):
            ctx.match(np.zeros((40000, 72), np.uint32), t[0])


@pytest.mark.gpu
def test_match_distance_is_32_bits_wide(sift):
    """DMatch_t::distance is an ap_uint<32> (SIFTParameters.hpp:124).  include/sift_hip.h supports rows whose
    distances stay below 2^32 (what compute_descriptor produces cannot come near); beyond that the sum wraps modulo
    2^32 like the field would.  This pins both sides of the documented limit, so that a later change cannot move it
    unnoticed."""
    with sift.SiftContext(64, 64) as ctx:
        q = np.zeros((2, 72), np.uint32)
        t = np.zeros((1, 72), np.uint32)
        q[0, 0] = q[0, 1] = 0x7FFFFFFF                                # distance 0xfffffffe to the zero row: the largest exact case
        m = ctx.match(q, t)
        assert int(m["distance"][0]) == 0xFFFFFFFE and m["trainIdx"][0] == 0
        q[0, 2] = 2                                                   # true distance 2^32: reads as 0 (wrapped), documented as unsupported
        m = ctx.match(q, t)
        assert int(m["distance"][0]) == 0 and m["trainIdx"][0] == 0
