"""Integer-only synthetic textured frames (SURVEY.md 8(d) generator).

h32(x): x^=x>>16; x*=0x7feb352d; x^=x>>15; x*=0x846ca68b; x^=x>>16   (uint32)
n_s(y,x) = h32(seed ^ h32((y>>s)*0x9E3779B1 + (x>>s)) ^ (s*0x85EBCA6B)) & 0xFF
pixel    = (4 n_5 + 3 n_4 + 2 n_3 + 2 n_2 + 1 n_0) / 12
"""
import numpy as np

_SCALES = ((5, 4), (4, 3), (3, 2), (2, 2), (0, 1))
_M = np.uint64(0xFFFFFFFF)


def _h32(x):
    x = x & _M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M
    x ^= x >> np.uint64(16)
    return x


def synth_frame(seed, width, height):
    """-> uint8 [height, width]"""
    y = np.arange(height, dtype=np.uint64)[:, None]
    x = np.arange(width, dtype=np.uint64)[None, :]
    acc = np.zeros((height, width), np.uint64)
    for s, w in _SCALES:
        s64 = np.uint64(s)
        inner = _h32(((y >> s64) * np.uint64(0x9E3779B1) + (x >> s64)) & _M)
        n = _h32((np.uint64(seed & 0xFFFFFFFF) ^ inner ^ np.uint64((s * 0x85EBCA6B) & 0xFFFFFFFF))) & np.uint64(0xFF)
        acc += np.uint64(w) * n
    return (acc // np.uint64(12)).astype(np.uint8)
