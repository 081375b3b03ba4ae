"""The C++ host layer above the C ABI: the reference's four entry points (include/HW_SIFT.hpp,
DetectKeypoints.hpp, SIFT_Descriptor.hpp) in libhw_sift.so, and the main.cpp-like driver."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

LIB = os.path.join(ROOT, "sift_amd", "lib", "libhw_sift.so")
MAIN = os.path.join(ROOT, "examples", "sift_main")
# C++ names with the reference's signatures (mangled: plain C++ functions, as in the reference)
SYMS = {
    "sift_detect_and_compute": "_Z23sift_detect_and_computePhP10Keypoint_tPji",
    "sift_detect": "_Z11sift_detectPhP6Grad_tP10Keypoint_ti",
    "compute_main_orientation": "_Z24compute_main_orientationP6Grad_tP10Keypoint_tPh",
    "compute_descriptor": "_Z18compute_descriptorP6Grad_tP10Keypoint_tPhPj",
    "sift_set_frame_size": "_Z19sift_set_frame_sizeii",
    "sift_select_final_keypoints": "_Z27sift_select_final_keypointsPK10Keypoint_tPS_Pj",
}


def test_host_library_exports_reference_entry_points():
    assert os.path.exists(LIB), "run __graft_entry__.build()"
    lib = C.CDLL(LIB)
    for name, sym in SYMS.items():
        assert hasattr(lib, sym), "%s (%s) missing from libhw_sift.so" % (name, sym)


def test_headers_compile_standalone(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "HW_SIFT.hpp"\n#include "DetectKeypoints.hpp"\n#include "SIFT_Descriptor.hpp"\n'
                   'static_assert(WIDTH == 640 && HEIGHT == 480 && MAX_KEYPOINTS == 10000 && DESC_SIZE == 72, "");\n'
                   'int main(){ void (*f)(Pixel_t*, Keypoint_t*, Desc_t*, int) = sift_detect_and_compute; return f == 0; }\n')
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-DWIDTH=640", "-DHEIGHT=480", "-c",
                           "-I", os.path.join(ROOT, "include"), "-o", str(tmp_path / "t.o"), str(src)])


@pytest.mark.gpu
def test_cpp_entry_points_match_oracle(sift, orc):
    lib = C.CDLL(LIB)
    W, H = 640, 480
    getattr(lib, SYMS["sift_set_frame_size"])(W, H)
    img = orc.synth_frame(1, W, H)
    kps = np.zeros(10000, sift.KP_DTYPE)
    desc = np.zeros((10000, 72), np.uint32)
    grad = np.zeros((H, W), sift.GRAD_DTYPE)
    ori = np.zeros(10000, np.uint8)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    getattr(lib, SYMS["sift_detect_and_compute"])(vp(img), vp(kps), vp(desc), 15)
    rk, rd, _ = orc.detect_and_compute(img)
    n = int((rk["contrast"] > 0).sum())
    assert (kps == rk).all() and (desc[:n] == rd[:n]).all() and not desc[n:].any()
    kps2 = np.zeros_like(kps)
    getattr(lib, SYMS["sift_detect"])(vp(img), vp(grad), vp(kps2), 15)
    ref = orc.sift_detect(img)
    assert (kps2 == rk).all() and (grad == ref["grad"]).all()
    getattr(lib, SYMS["compute_main_orientation"])(vp(grad), vp(kps2), vp(ori))
    assert (ori == orc.main_orientation(ref["grad"], rk)).all()
    d2 = np.zeros_like(desc)
    getattr(lib, SYMS["compute_descriptor"])(vp(grad), vp(kps2), vp(ori), vp(d2))
    assert (d2[:n] == rd[:n]).all()


@pytest.mark.gpu
def test_main_like_driver(tmp_path, orc):
    """main.cpp:37-210 flow: strongest MAX_FINAL_KEYPOINTS by contrast, descriptors by raster index"""
    out = subprocess.run([MAIN, "synth:640:480:1", str(tmp_path / "o")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "985 keypoints" in out.stdout
    rk, rd, _ = orc.detect_and_compute(orc.synth_frame(1, 640, 480))
    got = np.loadtxt(tmp_path / "o_keypoints.txt", dtype=np.int64)
    gd = np.loadtxt(tmp_path / "o_descriptors.txt", dtype=np.int64)
    assert got.shape == (985, 4) and gd.shape == (985, 72)
    assert (np.diff(got[:, 3]) <= 0).all()                       # sorted by contrast, descending
    idx = got[:, 0]
    assert sorted(idx.tolist()) == list(range(985))
    assert (got[:, 1] == rk["x"][idx]).all() and (got[:, 2] == rk["y"][idx]).all()
    assert (gd == rd[idx]).all()


@pytest.mark.gpu
def test_main_like_driver_reads_bmp(tmp_path, orc):
    """main.cpp:41-58: frames come as .bmp files read as gray; 8-bit (palette) and 24-bit files."""
    from PIL import Image
    W, H = 640, 480
    gray = orc.synth_frame(1, W, H)
    Image.fromarray(gray).save(tmp_path / "g8.bmp")                          # 8-bit, gray palette
    rng = np.random.default_rng(4)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "c24.bmp")                          # 24-bit, stored B, G, R bottom-up
    expect = {"g8": gray, "c24": orc.bgr2gray(rgb[..., ::-1].copy(), W)}
    for name, img in expect.items():
        out = subprocess.run([MAIN, str(tmp_path / (name + ".bmp")), str(tmp_path / name)], capture_output=True, text=True,
                             timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        rk, rd, _ = orc.detect_and_compute(img)
        n = int((rk["contrast"] > 0).sum())
        assert ("%d keypoints" % n) in out.stdout
        if n:
            got = np.loadtxt(tmp_path / (name + "_keypoints.txt"), dtype=np.int64, ndmin=2)
            idx = got[:, 0]
            assert len(idx) == min(n, 1000)
            assert (got[:, 1] == rk["x"][idx]).all() and (got[:, 2] == rk["y"][idx]).all()
