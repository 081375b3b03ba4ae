"""Frame partitioning across GPUs: frames are independent units, so a batch
of N frames is cut into contiguous ranges of ceil(N/G) per rank with no exchange step.  The
only collectives are the timing barrier and the max-over-ranks of the elapsed time; they run
on RCCL (backend "nccl") on GPUs and on gloo in the CPU tests."""
import os


def frame_range(nframes, rank, world):
    """-> (first, last) of the contiguous range of frames rank `rank` of `world` owns"""
    if world < 1 or not (0 <= rank < world) or nframes < 0:
        raise ValueError("bad partition (%d frames, rank %d of %d)" % (nframes, rank, world))
    per = (nframes + world - 1) // world
    lo = min(nframes, rank * per)
    return lo, min(nframes, lo + per)


def frame_seeds(first_seed, per_rank, rank):
    """weak scaling: rank r processes its own `per_rank` frames with distinct seeds"""
    return [first_seed + rank * per_rank + i for i in range(per_rank)]


def pick_device(local_rank, visible, share=False):
    """device index of the rank with LOCAL_RANK `local_rank` when `visible` devices are visible to it:
    its own when every rank sees the whole node, device 0 when a per-rank visibility mask leaves exactly
    one; `share` (the gloo rehearsal on a one-GPU box) lets several ranks use one device"""
    if visible < 1:
        raise RuntimeError("rank with LOCAL_RANK %d sees no GPU" % local_rank)
    if local_rank < visible:
        return local_rank
    if visible == 1 or share:
        return local_rank % visible
    raise RuntimeError("LOCAL_RANK %d but only %d GPUs visible" % (local_rank, visible))


def device_identity(props, index):
    """an int that names the physical GPU behind device `index` of this process: ranks that each see one
    device through a visibility mask all call it device 0, so the index alone cannot tell whether two
    ranks share a GPU.  From the device properties' uuid, else its PCI address, else the index."""
    uuid = getattr(props, "uuid", None)
    if uuid is not None:
        raw = getattr(uuid, "bytes", None)
        if raw is not None:                       # bytes, or a list of (possibly signed) byte values
            digits = bytes(int(v) & 0xFF for v in raw).hex()
        else:
            digits = "".join(ch for ch in str(uuid) if ch in "0123456789abcdefABCDEF")
        if digits and int(digits, 16):
            return int(digits, 16) & ((1 << 62) - 1)
    pci = [getattr(props, n, None) for n in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    if pci[1] is not None:
        return (1 << 40) | int(pci[0] or 0) << 16 | int(pci[1]) << 8 | int(pci[2] or 0)
    return int(index)


def check_ranks(world_expected, world_seen, identities, shared_ok):
    """what bench.py asserts before it times N > 1 ranks: the process group holds the N ranks asked for, and
    no two of them sit on one GPU (unless the gloo rehearsal on a one-GPU box asked for that)"""
    if world_seen != world_expected or len(identities) != world_expected:
        raise RuntimeError("asked for %d ranks, the process group has %d (%d answered)" % (world_expected, world_seen, len(identities)))
    if not shared_ok and len(set(identities)) != len(identities):
        raise RuntimeError("two ranks share a GPU: device identities %r" % (identities,))


class DistEnv:
    """One process per GPU, launched by torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""

    def __init__(self, backend=None, device=None, force_group=False):
        """force_group: build the process group even for one rank (the test that runs this class's collectives
        through RCCL on a one-GPU box)"""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = backend
        self._dist = None
        self._device = None
        if self.world > 1 or force_group:
            import torch
            import torch.distributed as dist
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            self.backend = backend
            if backend == "nccl":
                dev = self.local_rank if device is None else int(device)
                torch.cuda.set_device(dev)
                self._device = torch.device("cuda", dev)
            else:
                self._device = torch.device("cpu")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if not dist.is_initialized():
                if backend == "nccl":
                    dist.init_process_group(backend, device_id=self._device)
                else:
                    dist.init_process_group(backend)
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max_over_ranks(self, value):
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self._dist is None:
            return int(value)
        import torch
        t = torch.tensor([int(value)], dtype=torch.int64, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return int(t.item())

    def gather_ints(self, value):
        """every rank's `value` (an int below 2^63), in rank order"""
        if self._dist is None:
            return [int(value)]
        import torch
        mine = torch.tensor([int(value)], dtype=torch.int64, device=self._device)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self._dist.all_gather(out, mine)
        return [int(t.item()) for t in out]

    def dist_world_size(self):
        """the world size the process group itself reports (1 without one)"""
        return 1 if self._dist is None else int(self._dist.get_world_size())

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
        self._dist = None
