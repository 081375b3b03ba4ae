"""ctypes binding of the C ABI declared in include/sift_hip.h.

Only plain pointers and sizes cross this boundary.  Loading fails loudly when
the shared library is missing; nothing here falls back to a CPU path.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "libsift_hip.so")

MAX_KEYPOINTS = 10000   # SIFTParameters.hpp:22
DESC_SIZE = 72               # SIFTParameters.hpp:26
MAX_FINAL_KEYPOINTS = 1000   # SIFTParameters.hpp:24
NSTAGES = 5

# Keypoint_t (SIFTParameters.hpp:79-85) and Grad_t (:88-92)
KP_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("contrast", "<u4")])
GRAD_DTYPE = np.dtype([("ang", "<u2"), ("mag", "<u2")])
DMATCH_DTYPE = np.dtype([("distance", "<u4"), ("queryIdx", "<i2"), ("trainIdx", "<i2")])   # DMatch_t (:122-128)

# every symbol include/sift_hip.h declares (tests check the library exports all)
EXPORTS = (
    "sift_hip_create", "sift_hip_create_geometry", "sift_hip_destroy", "sift_hip_last_error", "sift_hip_device_count",
    "sift_hip_host_alloc", "sift_hip_host_free",
    "sift_hip_detect_and_compute", "sift_hip_detect", "sift_hip_main_orientation", "sift_hip_descriptor",
    "sift_hip_select_top", "sift_hip_select_top_dev", "sift_hip_gray_from_bgr", "sift_hip_gray_from_bgr_dev",
    "sift_hip_match", "sift_hip_match_dev", "sift_hip_batch", "sift_hip_batch_top", "sift_hip_batch_range", "sift_hip_debug_created", "sift_hip_detect_and_compute_dev", "sift_hip_detect_dev",
    "sift_hip_main_orientation_dev", "sift_hip_descriptor_dev", "sift_hip_sync", "sift_hip_debug_planes",
    "sift_hip_next_octave", "sift_hip_next_octave_dev",
    "sift_hip_set_profiling", "sift_hip_stage_ms", "sift_hip_stage_name", "sift_hip_set_blur_rows", "sift_hip_set_blur_mfma", "sift_hip_set_detect_rows", "sift_hip_set_detect_queue", "sift_hip_detect_and_compute_static", "sift_hip_static_scratch_needed", "sift_hip_set_graph_frames", "sift_hip_set_quad_sweep",
)


class SiftHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sift_hip error %d: %s" % (code, msg))
        self.code = code


def library_path():
    return _LIB


def build_library(force=False):
    """Compile sift_amd/csrc for gfx950 with hipcc (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", src, "clean"])
    subprocess.check_call(["make", "-s", "-C", src])
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "host")])
    return _LIB


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise SiftHipError(-3, "libsift_hip.so not built (%s); run sift_amd.build_library()" % _LIB)
    # A process that also uses PyTorch-ROCm must bring PyTorch's GPU runtime up first (INTEGRATION.md section 5:
    # the wheel carries its own HIP runtime, and torch.cuda does not initialise once /opt/rocm's is up).  When
    # torch is already imported that order is made here rather than left to the caller; torch is never imported
    # for it, and a process without torch is not affected.
    torch = sys.modules.get("torch")
    if torch is not None and getattr(getattr(torch, "version", None), "hip", None):
        try:
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:                                    # no usable GPU for torch: the library reports its own errors
            pass
    lib = C.CDLL(_LIB)
    vp, i32 = C.c_void_p, C.c_int
    lib.sift_hip_last_error.restype = C.c_char_p
    lib.sift_hip_stage_name.restype = C.c_char_p
    lib.sift_hip_stage_name.argtypes = [i32]
    lib.sift_hip_host_alloc.restype = vp
    lib.sift_hip_host_alloc.argtypes = [C.c_ulong]
    lib.sift_hip_host_free.restype = None
    lib.sift_hip_host_free.argtypes = [vp]
    lib.sift_hip_destroy.restype = None
    lib.sift_hip_destroy.argtypes = [vp]
    lib.sift_hip_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32]
    lib.sift_hip_create_geometry.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32, i32]
    lib.sift_hip_detect_and_compute.argtypes = [vp, vp, vp, vp, i32]
    lib.sift_hip_detect.argtypes = [vp, vp, vp, vp, i32]
    lib.sift_hip_main_orientation.argtypes = [vp, vp, vp, vp]
    lib.sift_hip_descriptor.argtypes = [vp, vp, vp, vp, vp]
    lib.sift_hip_batch.argtypes = [C.POINTER(vp), i32, C.POINTER(vp), i32, vp, vp, i32]
    lib.sift_hip_batch_top.argtypes = [C.POINTER(vp), i32, C.POINTER(vp), i32, i32, vp, vp, vp, vp, i32]
    lib.sift_hip_batch_range.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.sift_hip_debug_created.argtypes = [i32]
    lib.sift_hip_debug_created.restype = C.c_long
    lib.sift_hip_detect_and_compute_dev.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp]
    lib.sift_hip_detect_dev.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp]
    lib.sift_hip_main_orientation_dev.argtypes = [vp, vp, i32, vp, vp, vp]
    lib.sift_hip_descriptor_dev.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    lib.sift_hip_select_top.argtypes = [vp, vp, i32, vp, vp, C.POINTER(C.c_uint32)]
    lib.sift_hip_select_top_dev.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.sift_hip_gray_from_bgr.argtypes = [vp, vp, C.c_long, i32, vp]
    lib.sift_hip_gray_from_bgr_dev.argtypes = [vp, vp, i32, C.c_long, i32, vp, vp]
    lib.sift_hip_match.argtypes = [vp, vp, i32, vp, i32, vp]
    lib.sift_hip_match_dev.argtypes = [vp, vp, i32, vp, i32, i32, vp, vp]
    lib.sift_hip_sync.argtypes = [vp]
    lib.sift_hip_next_octave_dev.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.sift_hip_next_octave.argtypes = [vp, i32, i32, vp]
    lib.sift_hip_debug_planes.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    lib.sift_hip_set_profiling.argtypes = [vp, i32]
    lib.sift_hip_stage_ms.argtypes = [vp, i32, C.POINTER(C.c_float)]
    lib.sift_hip_set_blur_rows.argtypes = [vp, i32]
    lib.sift_hip_set_blur_mfma.argtypes = [vp, i32]
    lib.sift_hip_set_detect_rows.argtypes = [vp, i32]
    lib.sift_hip_set_detect_queue.argtypes = [vp, i32]
    lib.sift_hip_detect_and_compute_static.argtypes = [i32, vp, i32, i32, i32, vp, vp, vp, i32, vp]
    lib.sift_hip_static_scratch_needed.argtypes = [i32, i32, i32]
    lib.sift_hip_static_scratch_needed.restype = C.c_ulong
    lib.sift_hip_set_quad_sweep.argtypes = [vp, i32]
    lib.sift_hip_set_graph_frames.argtypes = [vp, i32]
    _lib = lib
    return lib


def check(rc):
    if rc < 0:
        raise SiftHipError(rc, load_library().sift_hip_last_error().decode("utf-8", "replace"))
    return rc


STAGE_NAMES = ("k_blur", "k_detect", "k_cull", "k_emit", "k_describe")
