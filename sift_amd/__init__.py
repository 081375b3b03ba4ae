"""sift_amd -- MI355X (gfx950) SIFT hot path behind the reference's entry points.

Host-side mirror of HW_SIFT / DetectKeypoints / SIFT_Descriptor over the C ABI
in include/sift_hip.h.  There is no CPU fallback: every compute entry point
needs libsift_hip.so and a gfx950 device and raises otherwise.
"""
from .capi import (DESC_SIZE, DMATCH_DTYPE, GRAD_DTYPE, KP_DTYPE, MAX_KEYPOINTS, STAGE_NAMES, SiftHipError, build_library,
                   library_path, load_library)
from .hw_sift import (SiftContext, compute_descriptor, compute_main_orientation, sift_detect,
                      sift_detect_and_compute, sift_detect_and_compute_octaves)
from .synth import synth_frame

__all__ = [
    "DESC_SIZE", "DMATCH_DTYPE", "GRAD_DTYPE", "KP_DTYPE", "MAX_KEYPOINTS", "STAGE_NAMES", "SiftHipError", "build_library",
    "library_path", "load_library", "SiftContext", "compute_descriptor", "compute_main_orientation",
    "sift_detect", "sift_detect_and_compute", "sift_detect_and_compute_octaves", "synth_frame",
]
