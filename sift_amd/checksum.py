"""Order-sensitive 64-bit checksum of an array of 32-bit words, the same value from numpy (host)
and from torch (device, int64 arithmetic that wraps modulo 2^64):

    sum over i of (word_i + 1) * (i * A + B)   mod 2^64

bench.py compares every frame's keypoint array and descriptor matrix with the committed per-seed
expectations through it (tests/golden/bench_expected.json) without moving 3 MB per frame to the host.
"""
import numpy as np

_A = 0x9E3779B97F4A7C15
_B = 0x632BE59BD9B4E019
_M = (1 << 64) - 1


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def words_checksum_np(words):
    """words: any contiguous array whose bytes are little-endian u32 words -> int in [0, 2^64)"""
    w = np.ascontiguousarray(words).view(np.uint32).reshape(-1).astype(np.uint64)
    with np.errstate(over="ignore"):
        i = np.arange(w.size, dtype=np.uint64)
        return int(((w + np.uint64(1)) * (i * np.uint64(_A) + np.uint64(_B))).sum(dtype=np.uint64)) & _M


def words_checksum_torch(t, frames_per_chunk=32):
    """t: torch int32 tensor [B, ...] on any device -> list of B ints in [0, 2^64), one per frame"""
    import torch
    B = t.shape[0]
    flat = t.reshape(B, -1)
    n = flat.shape[1]
    i = torch.arange(n, dtype=torch.int64, device=t.device)
    weight = i * _signed(_A) + _signed(_B)            # int64 multiply and add wrap
    out = []
    for b0 in range(0, B, frames_per_chunk):
        w = flat[b0:b0 + frames_per_chunk].to(torch.int64) & 0xFFFFFFFF
        s = ((w + 1) * weight).sum(dim=1)
        out.extend(int(v) & _M for v in s.cpu().tolist())
    return out
