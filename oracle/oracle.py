"""ctypes binding of the CPU oracle (oracle/sift_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under sift_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsift_oracle.so")

MAX_KEYPOINTS = 10000
DESC_SIZE = 72

KP_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("contrast", "<u4")])
GRAD_DTYPE = np.dtype([("ang", "<u2"), ("mag", "<u2")])


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "sift_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsift_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class _Planes(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("g1", "g2", "g3", "d1", "d2", "d3", "kp1", "kp2", "eratio")]


def gauss_table(which):
    n = {1: 5, 2: 11, 3: 19}[which]
    out = np.zeros((n, n), np.int32)
    lib().orc_gauss_table(C.c_int(which), _p(out))
    return out


def circle_mask():
    out = np.zeros(625, np.uint8)
    lib().orc_circle_mask(_p(out))
    return out.reshape(25, 25)


def mask_all():
    out = np.zeros(71 * 71, np.uint8)
    lib().orc_mask_all(_p(out))
    return out.reshape(71, 71)


def synth_frame(seed, W, H):
    out = np.zeros((H, W), np.uint8)
    lib().orc_synth_frame(C.c_uint32(seed), C.c_int(W), C.c_int(H), _p(out))
    return out


def blur(img):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    g = [np.zeros((H, W), np.uint8) for _ in range(3)]
    lib().orc_blur(_p(img), C.c_int(W), C.c_int(H), _p(g[0]), _p(g[1]), _p(g[2]))
    return g


def gradient(g2):
    g2 = np.ascontiguousarray(g2, np.uint8)
    H, W = g2.shape
    out = np.zeros((H, W), GRAD_DTYPE)
    lib().orc_gradient(_p(g2), C.c_int(W), C.c_int(H), _p(out))
    return out


def edge_ratio_q15(L, R, U, Dn, UL, UR, DL, DR, c):
    f = lib().orc_edge_ratio_q15
    f.restype = C.c_int32
    return f(*[C.c_int(int(v)) for v in (L, R, U, Dn, UL, UR, DL, DR, c)])


def sift_detect(img, contrast_th=15, max_kp=MAX_KEYPOINTS, planes=False, want_grad=True):
    """-> dict(kps, total, grad[, g1,g2,g3,d1,d2,d3,kp1,kp2,eratio])."""
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    kps = np.zeros(max_kp, KP_DTYPE)
    grad = np.zeros((H, W), GRAD_DTYPE) if want_grad else None
    res = {}
    keep = None
    if planes:
        res = dict(g1=np.zeros((H, W), np.uint8), g2=np.zeros((H, W), np.uint8), g3=np.zeros((H, W), np.uint8),
                   d1=np.zeros((H, W), np.int16), d2=np.zeros((H, W), np.int16), d3=np.zeros((H, W), np.int16),
                   kp1=np.zeros((H, W), np.uint16), kp2=np.zeros((H, W), np.uint16),
                   eratio=np.zeros((H, W), np.int32))
        keep = _Planes(**{k: v.ctypes.data for k, v in res.items()})
    f = lib().orc_sift_detect_planes
    f.restype = C.c_int
    total = f(_p(img), C.c_int(W), C.c_int(H), _p(grad) if want_grad else None, _p(kps),
              C.c_int(contrast_th), C.c_int(max_kp), C.byref(keep) if keep is not None else None)
    res.update(kps=kps, total=total, grad=grad)
    return res


def main_orientation(grad, kps, geometry=0):
    """geometry 0 = Cam0 (the reference's code), 1 = Cam1 (its tables, semantics defined in sift_oracle.c)"""
    grad = np.ascontiguousarray(grad)
    H, W = grad.shape
    kps = np.ascontiguousarray(kps)
    ori = np.zeros(len(kps), np.uint8)
    lib().orc_main_orientation_geo(_p(grad), C.c_int(W), C.c_int(H), _p(kps), C.c_int(len(kps)), _p(ori), C.c_int(geometry))
    return ori


def descriptor(grad, kps, ori, geometry=0):
    grad = np.ascontiguousarray(grad)
    H, W = grad.shape
    kps = np.ascontiguousarray(kps)
    ori = np.ascontiguousarray(ori, np.uint8)
    desc = np.zeros((len(kps), DESC_SIZE), np.uint32)
    lib().orc_descriptor_geo(_p(grad), C.c_int(W), C.c_int(H), _p(kps), _p(ori), C.c_int(len(kps)), _p(desc), C.c_int(geometry))
    return desc


def detect_and_compute(img, contrast_th=15, geometry=0):
    """-> (kps[10000], desc[10000,72], total)."""
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    kps = np.zeros(MAX_KEYPOINTS, KP_DTYPE)
    desc = np.zeros((MAX_KEYPOINTS, DESC_SIZE), np.uint32)
    f = lib().orc_detect_and_compute_geo
    f.restype = C.c_int
    total = f(_p(img), C.c_int(W), C.c_int(H), _p(kps), _p(desc), C.c_int(contrast_th), C.c_int(geometry))
    return kps, desc, total


def next_octave(img, plane=2):
    """the build's definition of a further octave (include/sift_hip.h; the reference runs one octave, SURVEY.md
    8(f)4): pixel (2x, 2y) of blurred plane `plane` of `img` -> uint8[H // 2, W // 2]"""
    H, W = np.asarray(img).shape
    g = blur(img)[plane - 1]
    return np.ascontiguousarray(g[0:2 * (H // 2):2, 0:2 * (W // 2):2])


def detect_and_compute_octaves(img, noctaves, contrast_th=15, plane=2):
    """[(kps, desc, total)] per octave: detect_and_compute on img, then on next_octave of it, ..."""
    out = []
    for o in range(noctaves):
        out.append(detect_and_compute(img, contrast_th))
        if o + 1 < noctaves:
            img = next_octave(img, plane)
    return out


def geometry_masks(geometry):
    """-> (window, radius, circle mask [m, m], xstart[9], ystart[9], orientation mask [window, window])"""
    win, rad, m = ((71, 35, 25), (81, 40, 29))[geometry]
    cm = np.zeros(m * m, np.uint8)
    lib().orc_circle_mask_geo(C.c_int(geometry), _p(cm))
    ma = np.zeros(win * win, np.uint8)
    lib().orc_mask_all_geo(C.c_int(geometry), _p(ma))
    xs = ((45, 32, 14, 1, 1, 14, 32, 45, 23), (52, 37, 15, 0, 0, 15, 37, 52, 26))[geometry]
    ys = ((32, 45, 45, 32, 14, 1, 1, 14, 23), (37, 52, 52, 37, 15, 0, 0, 15, 26))[geometry]
    return win, rad, cm.reshape(m, m), xs, ys, ma.reshape(win, win)


def fnv1a(buf):
    """64-bit FNV-1a of a contiguous numpy array / bytes, as a 16-digit hex string."""
    a = np.frombuffer(buf, np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    f = lib().orc_fnv1a
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_size_t]
    return "%016x" % f(a.ctypes.data, a.size)


def select_top(kps, nfinal=1000):
    """main.cpp:82-130 -> (final[nfinal], index[nfinal], nvalid); ties ordered by slot."""
    kps = np.ascontiguousarray(kps)
    final = np.zeros(nfinal, KP_DTYPE)
    index = np.zeros(nfinal, np.uint32)
    f = lib().orc_select_top
    f.restype = C.c_int
    nvalid = f(_p(kps), C.c_int(len(kps)), C.c_int(nfinal), _p(final), _p(index))
    return final, index, nvalid


def bgr2gray(bgr, W, bottom_up=False):
    """main.cpp:51-58 for a colour frame: bgr [H, pitch] uint8 -> gray [H, W] (OpenCV's fixed-point weights)."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    if bgr.ndim == 3:
        bgr = bgr.reshape(bgr.shape[0], -1)
    H, pitch = bgr.shape
    gray = np.zeros((H, W), np.uint8)
    lib().orc_bgr2gray(_p(bgr), C.c_int(W), C.c_int(H), C.c_long(pitch), C.c_int(int(bool(bottom_up))), _p(gray))
    return gray


DMATCH_DTYPE = np.dtype([("distance", "<u4"), ("queryIdx", "<i2"), ("trainIdx", "<i2")])


def match(query_desc, train_desc):
    """DMatch_t per query row: nearest train row in the L1 sense (defined semantics; the reference has the type only)"""
    q = np.ascontiguousarray(query_desc, np.uint32).reshape(-1, DESC_SIZE)
    t = np.ascontiguousarray(train_desc, np.uint32).reshape(-1, DESC_SIZE)
    out = np.zeros(len(q), DMATCH_DTYPE)
    lib().orc_match(_p(q), C.c_int(len(q)), _p(t), C.c_int(len(t)), _p(out))
    return out
