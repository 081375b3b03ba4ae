"""The C-ABI shared library: loads, exports every symbol include/sift_hip.h declares, and
fails loudly (no CPU fallback) where there is no gfx950 device.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    with open(os.path.join(ROOT, "include", "sift_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sift_hip_[a-z_0-9]+)\s*\(", text)))


def test_header_compiles_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "sift_hip.h"\nint main(void){return sizeof(sift_keypoint)==8 && sizeof(sift_grad)==4 ? 0 : 1;}\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        exe = tmp_path / ("t_" + cc)
        args = [cc, std, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe)]
        args += ["-x", "c++" if cc == "g++" else "c", str(src)]
        subprocess.check_call(args)
        assert subprocess.call([str(exe)]) == 0


def test_library_exports_every_declared_symbol(sift):
    lib = sift.load_library()
    names = declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), "libsift_hip.so does not export " + n
    assert set(names) == set(sift.capi.EXPORTS)


def test_library_has_gfx950_code_object(sift):
    with open(sift.library_path(), "rb") as f:
        blob = f.read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_"):
        assert b"amdgcn-amd-amdhsa--" + other not in blob


def test_argument_errors_without_touching_a_device(sift):
    lib = sift.load_library()
    h = C.c_void_p()
    assert lib.sift_hip_create(None, 0, 640, 480, 10000, 1) == -1
    assert lib.sift_hip_create(C.byref(h), 0, 0, 480, 10000, 1) == -1
    assert b"width" in lib.sift_hip_last_error()
    assert lib.sift_hip_create(C.byref(h), 0, 70000, 480, 10000, 1) == -1
    assert lib.sift_hip_create_geometry(C.byref(h), 0, 640, 480, 10000, 1, 7) == -1
    assert b"geometry" in lib.sift_hip_last_error()
    assert lib.sift_hip_create(C.byref(h), 0, 640, 0, 10000, 1) == -1
    assert lib.sift_hip_create(C.byref(h), 0, 640, 480, 0, 1) == -1
    assert lib.sift_hip_create(C.byref(h), 0, 640, 480, 10000, 0) == -1
    assert lib.sift_hip_detect_and_compute(None, None, None, None, 15) == -1
    assert lib.sift_hip_sync(None) == -1
    assert lib.sift_hip_stage_name(0) == b"k_blur" and lib.sift_hip_stage_name(99) == b""
    lib.sift_hip_destroy(None)                                                  # no-op


def test_no_cpu_fallback_without_device(sift):
    """Where no HIP device exists (this container) every compute path must refuse to run."""
    lib = sift.load_library()
    if lib.sift_hip_device_count() > 0:
        pytest.skip("a HIP device is present; covered by the gpu tests")
    h = C.c_void_p()
    assert lib.sift_hip_create(C.byref(h), 0, 640, 480, 10000, 1) == -3         # SIFT_HIP_ENODEV
    assert not h.value
    with pytest.raises(sift.SiftHipError) as ei:
        sift.SiftContext(640, 480)
    assert ei.value.code == -3
    with pytest.raises(sift.SiftHipError):
        sift.sift_detect_and_compute(np.zeros((480, 640), np.uint8))


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under sift_amd/ or include/ may reference it."""
    bad = []
    for base in ("sift_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for fn in fs:
                if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")) or fn == "Makefile":
                    with open(os.path.join(dp, fn), errors="replace") as f:
                        t = f.read()
                    if re.search(r"\boracle\b|sift_oracle|liborc|orc_", t):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


@pytest.mark.gpu
def test_loader_brings_torch_up_first():
    """a process that imported torch without touching the GPU, then uses this library, then
    torch.cuda -- the loader initialises torch's runtime first, so all three work (a fresh process: the order matters
    once per process)."""
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import sift_amd as sift\n"
        "with sift.SiftContext(256, 192) as ctx:\n"
        "    kps, desc = ctx.sift_detect_and_compute(np.zeros((192, 256), np.uint8), 15)\n"
        "assert not np.asarray(desc).any()\n"
        "t = torch.arange(8, device='cuda:0')\n"
        "assert int(t.sum().item()) == 28\n"
        "print('ok')\n" % ROOT)
    r = subprocess.run([os.environ.get("PYTHON", "python"), "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
