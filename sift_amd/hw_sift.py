"""Host-side mirror of the reference's entry points over the C ABI.

Names and argument meaning follow HW_SIFT.hpp:11-15, DetectKeypoints.hpp:109-113
and SIFT_Descriptor.hpp:12-14/:59-62; frame size is a property of the context
instead of the compile-time WIDTH/HEIGHT of SIFTParameters.hpp:17-18.  Outputs
have the reference's fixed shapes: MAX_KEYPOINTS keypoint slots (zero padded)
and MAX_KEYPOINTS x 72 descriptor words.
"""
import sys as _sys
import ctypes as C

import numpy as np

from . import capi
from .capi import (DESC_SIZE, DMATCH_DTYPE, GRAD_DTYPE, KP_DTYPE, MAX_FINAL_KEYPOINTS, MAX_KEYPOINTS, NSTAGES, check,
                   load_library)


def _vp(a):
    return C.c_void_p(a.ctypes.data)


def _dp(p):
    """device pointer (int, or anything with data_ptr()) -> c_void_p"""
    if p is None:
        return None
    if hasattr(p, "data_ptr"):
        p = p.data_ptr()
    return C.c_void_p(int(p))


class SiftContext:
    """One device + stream + persistent scratch (sift_hip_create)."""

    def __init__(self, width, height, device=0, max_keypoints=MAX_KEYPOINTS, max_batch=1, geometry=0):
        """geometry: 0 = Cam0 (the reference's code), 1 = Cam1 (its second set of descriptor tables)"""
        self._lib = load_library()
        self.width, self.height = int(width), int(height)
        self.max_keypoints, self.max_batch, self.device = int(max_keypoints), int(max_batch), int(device)
        self.geometry = int(geometry)
        h = C.c_void_p()
        check(self._lib.sift_hip_create_geometry(C.byref(h), self.device, self.width, self.height, self.max_keypoints,
                                                 self.max_batch, self.geometry))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sift_hip_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        # prefer close() or the context manager; at interpreter shutdown the HIP runtime may already be
        # torn down, so nothing is destroyed then (the process is about to release everything anyway)
        try:
            if _sys is None or _sys.is_finalizing():     # module globals are cleared at shutdown: no import here
                return
            self.close()
        except Exception:
            pass

    # -- host-buffer entry points -------------------------------------------------
    def _img(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        if img.shape != (self.height, self.width):
            raise ValueError("image must be %dx%d uint8, got %r" % (self.height, self.width, img.shape))
        return img

    def _grad(self, grad):
        grad = np.ascontiguousarray(grad, GRAD_DTYPE)
        if grad.shape != (self.height, self.width):
            raise ValueError("gradient plane must be %dx%d Grad_t" % (self.height, self.width))
        return grad

    def _kps(self, kps):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        if kps.shape != (self.max_keypoints,):
            raise ValueError("keypoint array must have %d slots" % self.max_keypoints)
        return kps

    def sift_detect_and_compute(self, img, contrastTh=15):
        img = self._img(img)
        kps = np.empty(self.max_keypoints, KP_DTYPE)
        desc = np.empty((self.max_keypoints, DESC_SIZE), np.uint32)
        check(self._lib.sift_hip_detect_and_compute(self._h, _vp(img), _vp(kps), _vp(desc), int(contrastTh)))
        return kps, desc

    def sift_detect(self, img, contrastTh=15, want_grad=True):
        img = self._img(img)
        kps = np.empty(self.max_keypoints, KP_DTYPE)
        grad = np.empty((self.height, self.width), GRAD_DTYPE) if want_grad else None
        check(self._lib.sift_hip_detect(self._h, _vp(img), _vp(grad) if want_grad else None, _vp(kps),
                                        int(contrastTh)))
        return grad, kps

    def compute_main_orientation(self, grad, kps):
        grad, kps = self._grad(grad), self._kps(kps)
        ori = np.empty(self.max_keypoints, np.uint8)
        check(self._lib.sift_hip_main_orientation(self._h, _vp(grad), _vp(kps), _vp(ori)))
        return ori

    def compute_descriptor(self, grad, kps, ori):
        grad, kps = self._grad(grad), self._kps(kps)
        ori = np.ascontiguousarray(ori, np.uint8)
        if ori.shape != (self.max_keypoints,):
            raise ValueError("orientation array must have %d entries" % self.max_keypoints)
        desc = np.empty((self.max_keypoints, DESC_SIZE), np.uint32)
        check(self._lib.sift_hip_descriptor(self._h, _vp(grad), _vp(kps), _vp(ori), _vp(desc)))
        return desc

    def gray_from_bgr(self, bgr, bottom_up=False):
        """main.cpp:51-58 for a colour frame: bgr [H, pitch] uint8 rows (pitch >= 3*W, multiple of 4) -> gray [H, W]"""
        bgr = np.ascontiguousarray(bgr, np.uint8)
        if bgr.ndim == 3:
            bgr = bgr.reshape(bgr.shape[0], -1)
        if bgr.shape[0] != self.height:
            raise ValueError("frame must have %d rows" % self.height)
        gray = np.empty((self.height, self.width), np.uint8)
        check(self._lib.sift_hip_gray_from_bgr(self._h, _vp(bgr), int(bgr.shape[1]), int(bool(bottom_up)), _vp(gray)))
        return gray

    def gray_from_bgr_dev(self, d_bgr, nframes, pitch, bottom_up, d_gray, stream=None):
        check(self._lib.sift_hip_gray_from_bgr_dev(self._h, _dp(d_bgr), int(nframes), int(pitch), int(bool(bottom_up)),
                                                   _dp(d_gray), _dp(stream)))

    def select_top(self, kps, nfinal=MAX_FINAL_KEYPOINTS):
        """main.cpp:82-130 -> (pFinalKeypoints[nfinal], slot index[nfinal], nKp); equal contrasts keep slot order"""
        kps = self._kps(kps)
        final = np.empty(int(nfinal), KP_DTYPE)
        index = np.empty(int(nfinal), np.uint32)
        nvalid = C.c_uint32(0)
        check(self._lib.sift_hip_select_top(self._h, _vp(kps), int(nfinal), _vp(final), _vp(index), C.byref(nvalid)))
        return final, index, int(nvalid.value)

    def match(self, query_desc, train_desc):
        """DMatch_t[nquery] of the nearest train row in the L1 sense (the reference has the type only)"""
        q = np.ascontiguousarray(query_desc, np.uint32).reshape(-1, DESC_SIZE)
        t = np.ascontiguousarray(train_desc, np.uint32).reshape(-1, DESC_SIZE)
        m = np.empty(len(q), DMATCH_DTYPE)
        check(self._lib.sift_hip_match(self._h, _vp(q), len(q), _vp(t), len(t), _vp(m)))
        return m

    def match_dev(self, d_query, nquery, d_train, ntrain, npairs, d_matches, stream=None):
        check(self._lib.sift_hip_match_dev(self._h, _dp(d_query), int(nquery), _dp(d_train), int(ntrain), int(npairs),
                                           _dp(d_matches), _dp(stream)))

    # -- device-buffer entry points (pointers into this device's memory) -----------
    def select_top_dev(self, d_kps, nframes, nfinal, d_final, d_index=None, d_nvalid=None, d_desc=None,
                       d_final_desc=None, stream=None):
        check(self._lib.sift_hip_select_top_dev(self._h, _dp(d_kps), int(nframes), int(nfinal), _dp(d_final),
                                                _dp(d_index), _dp(d_nvalid), _dp(d_desc), _dp(d_final_desc),
                                                _dp(stream)))

    def detect_and_compute_dev(self, d_imgs, nframes, d_kps, d_desc, d_total=None, contrastTh=15, stream=None):
        check(self._lib.sift_hip_detect_and_compute_dev(self._h, _dp(d_imgs), int(nframes), _dp(d_kps), _dp(d_desc),
                                                        _dp(d_total), int(contrastTh), _dp(stream)))

    def detect_dev(self, d_imgs, nframes, d_grad, d_kps, d_total=None, contrastTh=15, stream=None):
        check(self._lib.sift_hip_detect_dev(self._h, _dp(d_imgs), int(nframes), _dp(d_grad), _dp(d_kps),
                                            _dp(d_total), int(contrastTh), _dp(stream)))

    def main_orientation_dev(self, d_grad, nframes, d_kps, d_ori, stream=None):
        check(self._lib.sift_hip_main_orientation_dev(self._h, _dp(d_grad), int(nframes), _dp(d_kps), _dp(d_ori),
                                                      _dp(stream)))

    def descriptor_dev(self, d_grad, nframes, d_kps, d_ori, d_desc, stream=None):
        check(self._lib.sift_hip_descriptor_dev(self._h, _dp(d_grad), int(nframes), _dp(d_kps), _dp(d_ori),
                                                _dp(d_desc), _dp(stream)))

    def sync(self):
        check(self._lib.sift_hip_sync(self._h))

    # -- introspection ------------------------------------------------------------
    def next_octave(self, frame=0, plane=2):
        """extension: the next octave's input frame, pixel (2x, 2y) of blurred plane `plane` (1, 2 or 3) of frame
        `frame` of the last detect call -> uint8[height // 2, width // 2]"""
        out = np.empty((self.height // 2, self.width // 2), np.uint8)
        check(self._lib.sift_hip_next_octave(self._h, int(frame), int(plane), _vp(out)))
        return out

    def next_octave_dev(self, frame_first, nframes, d_dst, plane=2, stream=None):
        check(self._lib.sift_hip_next_octave_dev(self._h, int(frame_first), int(nframes), int(plane), _dp(d_dst), _dp(stream)))

    def debug_planes(self, frame=0):
        H, W = self.height, self.width
        out = dict(g1=np.empty((H, W), np.uint8), g2=np.empty((H, W), np.uint8), g3=np.empty((H, W), np.uint8),
                   cand=np.empty((H, W), np.uint16), eratio=np.empty((H, W), np.int32),
                   surv=np.empty((H, W), np.uint16))
        check(self._lib.sift_hip_debug_planes(self._h, int(frame), _vp(out["g1"]), _vp(out["g2"]), _vp(out["g3"]),
                                              _vp(out["cand"]), _vp(out["eratio"]), _vp(out["surv"])))
        return out

    def set_profiling(self, nslots=1):
        """keep HIP-event kernel times of the last `nslots` calls (0 = off)"""
        check(self._lib.sift_hip_set_profiling(self._h, int(nslots)))

    def set_blur_rows(self, rows):
        check(self._lib.sift_hip_set_blur_rows(self._h, int(rows)))

    def set_blur_mfma(self, on):
        """comparison knob: False runs the blurs on the vector ALUs (k_blur) instead of the matrix cores"""
        check(self._lib.sift_hip_set_blur_mfma(self._h, 1 if on else 0))

    def set_graph_frames(self, max_frames):
        check(self._lib.sift_hip_set_graph_frames(self._h, int(max_frames)))

    def set_quad_sweep(self, on):
        """test knob: False sends every descriptor window through the pixel-by-pixel sweep"""
        check(self._lib.sift_hip_set_quad_sweep(self._h, 1 if on else 0))

    def set_detect_rows(self, rows):
        check(self._lib.sift_hip_set_detect_rows(self._h, int(rows)))

    def set_detect_queue(self, entries):
        """test knob: LDS queue entries k_detect fills before draining mid-strip (0 = default)"""
        check(self._lib.sift_hip_set_detect_queue(self._h, int(entries)))

    def stage_ms(self, back=0):
        """kernel times (ms) of the call `back` calls before the most recent one"""
        ms = (C.c_float * NSTAGES)()
        check(self._lib.sift_hip_stage_ms(self._h, int(back), ms))
        return dict(zip(capi.STAGE_NAMES, [float(v) for v in ms]))

    def stage_ms_all(self):
        """list of per-call dicts for every call the profiling ring holds, oldest first"""
        ms = (C.c_float * NSTAGES)()
        n = check(self._lib.sift_hip_stage_ms(self._h, 0, ms))
        return [self.stage_ms(b) for b in range(n - 1, -1, -1)]


def batch(contexts, imgs, contrastTh=15):
    """sift_hip_batch: context i takes the contiguous frame range batch_range(n, len(contexts), i);
    returns (kps[n,K], desc[n,K,72])."""
    lib = load_library()
    c0 = contexts[0]
    imgs = [c0._img(im) for im in imgs]
    n, K = len(imgs), c0.max_keypoints
    kps = np.empty((n, K), KP_DTYPE)
    desc = np.empty((n, K, DESC_SIZE), np.uint32)
    hs = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    ps = (C.c_void_p * max(n, 1))(*[im.ctypes.data for im in imgs])
    check(lib.sift_hip_batch(hs, len(contexts), ps, n, _vp(kps), _vp(desc), int(contrastTh)))
    return kps, desc


def batch_range(nframes, nctx, i):
    """the partition rule of sift_hip_batch (needs no device): (first, last) of context i's frames"""
    a, b = C.c_int(0), C.c_int(0)
    check(load_library().sift_hip_batch_range(int(nframes), int(nctx), int(i), C.byref(a), C.byref(b)))
    return a.value, b.value


def created_counts():
    """(streams, events, device buffers) the library has created in this process so far"""
    lib = load_library()
    return tuple(int(lib.sift_hip_debug_created(k)) for k in range(3))


def batch_top(contexts, imgs, nfinal=MAX_FINAL_KEYPOINTS, contrastTh=15):
    """sift_hip_batch_top: the image loop with main.cpp:82-130 on the device; returns
    (final_kps[n,nfinal], slot[n,nfinal], final_desc[n,nfinal,72], nKp[n])."""
    lib = load_library()
    c0 = contexts[0]
    imgs = [c0._img(im) for im in imgs]
    n, NF = len(imgs), int(nfinal)
    fin = np.empty((n, NF), KP_DTYPE)
    idx = np.empty((n, NF), np.uint32)
    fdesc = np.empty((n, NF, DESC_SIZE), np.uint32)
    nv = np.empty(n, np.uint32)
    hs = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    ps = (C.c_void_p * max(n, 1))(*[im.ctypes.data for im in imgs])
    check(lib.sift_hip_batch_top(hs, len(contexts), ps, n, NF, _vp(fin), _vp(idx), _vp(fdesc), _vp(nv), int(contrastTh)))
    return fin, idx, fdesc, nv


# ---- module-level functions with the reference's names ---------------------------
_default_ctx = {}


def _ctx_for(shape, device=0):
    key = (int(shape[0]), int(shape[1]), int(device))
    if key not in _default_ctx:
        _default_ctx[key] = SiftContext(key[1], key[0], device=key[2])
    return _default_ctx[key]


def sift_detect_and_compute(pSrcImage, contrastTh=15):
    """HW_SIFT.cpp:16-50 -> (pKeypoints[10000], pDesc[10000,72])"""
    pSrcImage = np.asarray(pSrcImage)
    return _ctx_for(pSrcImage.shape).sift_detect_and_compute(pSrcImage, contrastTh)


def sift_detect_and_compute_octaves(pSrcImage, noctaves, contrastTh=15, plane=2):
    """extension (include/sift_hip.h, "further octaves"): [(pKeypoints[10000], pDesc[10000,72])] per octave; octave
    o + 1 is pixel (2x, 2y) of blurred plane `plane` of octave o, coordinates are in the octave's own pixels"""
    img = np.asarray(pSrcImage)
    out = []
    for o in range(int(noctaves)):
        ctx = _ctx_for(img.shape)
        out.append(ctx.sift_detect_and_compute(img, contrastTh))
        if o + 1 < int(noctaves):
            img = ctx.next_octave(0, plane)
    return out


def sift_detect(pSrcImage, contrastTh=15):
    """DetectKeypoints.cpp:1073-1175 -> (pGradients[H,W], pKeypoints[10000])"""
    pSrcImage = np.asarray(pSrcImage)
    return _ctx_for(pSrcImage.shape).sift_detect(pSrcImage, contrastTh)


def compute_main_orientation(pGrad, pKeypoint):
    """SIFT_Descriptor.cpp:18-103 -> pMainOrientation[10000]"""
    return _ctx_for(np.asarray(pGrad).shape).compute_main_orientation(pGrad, pKeypoint)


def compute_descriptor(pGrad, pKeypoint, pMainOrientation):
    """SIFT_Descriptor.cpp:419-478 -> pDescriptorOut[10000,72]"""
    return _ctx_for(np.asarray(pGrad).shape).compute_descriptor(pGrad, pKeypoint, pMainOrientation)
