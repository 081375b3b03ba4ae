"""Integer-only synthetic textured frames (SURVEY.md 8(d) generator).

h32(x): x^=x>>16; x*=0x7feb352d; x^=x>>15; x*=0x846ca68b; x^=x>>16   (uint32)
n_s(y,x) = h32(seed ^ h32((y>>s)*0x9E3779B1 + (x>>s)) ^ (s*0x85EBCA6B)) & 0xFF
pixel    = (4 n_5 + 3 n_4 + 2 n_3 + 2 n_2 + 1 n_0) / 12
"""
import numpy as np

_SCALES = ((5, 4), (4, 3), (3, 2), (2, 2), (0, 1))


def _h32(x):
    # uint32 array arithmetic wraps modulo 2^32, which is what the generator specifies
    x = x ^ (x >> np.uint32(16))
    x = x * np.uint32(0x7FEB352D)
    x = x ^ (x >> np.uint32(15))
    x = x * np.uint32(0x846CA68B)
    x = x ^ (x >> np.uint32(16))
    return x


def synth_frame(seed, width, height):
    """-> uint8 [height, width]"""
    acc = np.zeros((height, width), np.uint32)
    seed32 = np.uint32(seed & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for s, w in _SCALES:
            # the noise is constant on 2^s x 2^s cells: evaluate per cell, then repeat
            cy = np.arange((height + (1 << s) - 1) >> s, dtype=np.uint32)[:, None]
            cx = np.arange((width + (1 << s) - 1) >> s, dtype=np.uint32)[None, :]
            inner = _h32(cy * np.uint32(0x9E3779B1) + cx)
            n = _h32(seed32 ^ inner ^ np.uint32((s * 0x85EBCA6B) & 0xFFFFFFFF)) & np.uint32(0xFF)
            if s:
                n = np.repeat(np.repeat(n, 1 << s, axis=0), 1 << s, axis=1)[:height, :width]
            acc += np.uint32(w) * n
    return (acc // np.uint32(12)).astype(np.uint8)


def synth_frames_torch(seeds, width, height, device="cpu"):
    """The same frames as synth_frame, for a list of seeds, computed with torch on `device`
    (int64 arithmetic masked to 32 bits) -> uint8 tensor [len(seeds), height, width]."""
    import torch
    M = 0xFFFFFFFF

    def h32(x):
        x = x ^ (x >> 16)
        x = (x * 0x7FEB352D) & M
        x = x ^ (x >> 15)
        x = (x * 0x846CA68B) & M          # the product may wrap past 2^63: the low 32 bits are still right
        x = x ^ (x >> 16)
        return x

    seed = torch.tensor([int(s) & M for s in seeds], dtype=torch.int64, device=device)[:, None, None]
    acc = torch.zeros((len(seeds), height, width), dtype=torch.int64, device=device)
    for s, w in _SCALES:
        cy = torch.arange((height + (1 << s) - 1) >> s, dtype=torch.int64, device=device)[:, None]
        cx = torch.arange((width + (1 << s) - 1) >> s, dtype=torch.int64, device=device)[None, :]
        inner = h32(((cy * 0x9E3779B1) & M) + cx & M)
        n = h32(seed ^ inner[None] ^ ((s * 0x85EBCA6B) & M)) & 0xFF
        if s:
            n = n.repeat_interleave(1 << s, dim=1).repeat_interleave(1 << s, dim=2)[:, :height, :width]
        acc += w * n
    return (acc // 12).to(torch.uint8)
