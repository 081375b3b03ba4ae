"""Frame partitioning and the N>1 timing protocol, world_size 2 on gloo (CPU)."""
import os
import socket

import pytest
import torch.multiprocessing as mp

from sift_amd.shard import DistEnv, frame_range, frame_seeds


def test_frame_ranges_cover_every_frame_once():
    for n in (0, 1, 7, 8, 40, 41, 1000):
        for g in (1, 2, 3, 4, 8):
            seen = []
            for r in range(g):
                lo, hi = frame_range(n, r, g)
                assert 0 <= lo <= hi <= n and hi - lo <= (n + g - 1) // g
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    with pytest.raises(ValueError):
        frame_range(10, 2, 2)


def test_library_partitions_like_frame_range():
    """sift_hip_batch's rule (sift_hip_batch_range, no device needed) is shard.frame_range: SURVEY.md 8(e)"""
    from sift_amd.hw_sift import batch_range
    from sift_amd.capi import SiftHipError
    for n in (0, 1, 7, 8, 11, 40, 41, 1000):
        for g in (1, 2, 3, 4, 8):
            for r in range(g):
                assert batch_range(n, g, r) == frame_range(n, r, g)
    with pytest.raises(SiftHipError):
        batch_range(10, 2, 2)


def test_rank_and_device_checks():
    """what bench.py asserts before timing N > 1 ranks (the line must prove RCCL saw N ranks)"""
    from types import SimpleNamespace as NS
    from sift_amd.shard import check_ranks, device_identity
    check_ranks(1, 1, [5], False)
    check_ranks(4, 4, [1, 2, 3, 4], False)
    check_ranks(2, 2, [7, 7], True)                                  # the gloo rehearsal on a one-GPU box
    for bad in ((4, 2, [1, 2], False), (2, 2, [7, 7], False), (2, 2, [7], False)):
        with pytest.raises(RuntimeError):
            check_ranks(*bad)
    a = device_identity(NS(uuid=NS(bytes=bytes(range(1, 17)))), 0)
    b = device_identity(NS(uuid=NS(bytes=bytes(range(2, 18)))), 0)
    assert a != b and 0 < a < 2 ** 63 and 0 < b < 2 ** 63              # two masked ranks both call their GPU "device 0"
    assert device_identity(NS(uuid="GPU-0123abcd"), 3) == 0x0123abcd
    assert device_identity(NS(uuid=NS(bytes=[0, -1, 3])), 0) == 0x00ff03     # torch hands the bytes over as a list of ints
    assert device_identity(NS(uuid=NS(bytes=bytes(16)), pci_domain_id=0, pci_bus_id=0x65, pci_device_id=0), 0) != \
        device_identity(NS(uuid=NS(bytes=bytes(16)), pci_domain_id=0, pci_bus_id=0x75, pci_device_id=0), 0)
    assert device_identity(NS(), 5) == 5


def test_frame_seeds_disjoint():
    s = [frame_seeds(1, 16, r) for r in range(8)]
    flat = sum(s, [])
    assert len(set(flat)) == 128 and flat[0] == 1


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env = DistEnv(backend="gloo")
    lo, hi = frame_range(41, env.rank, env.world)
    env.barrier()
    t = env.max_over_ranks(1.0 + rank)          # the slowest rank sets the job time
    total = env.sum_over_ranks(hi - lo)
    assert env.gather_ints(100 + rank) == [100, 101] and env.dist_world_size() == 2
    q.put((rank, t, total))
    env.close()


def test_two_rank_protocol_on_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert out == [(0, 2.0, 41), (1, 2.0, 41)]
