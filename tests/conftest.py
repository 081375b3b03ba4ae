"""pytest configuration: the `gpu` marker and shared fixtures.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI exports (no compute).
`-m gpu`      : HIP path vs oracle through the C ABI (needs a gfx950 device).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def fnv1a(buf):
    """64-bit FNV-1a of a bytes-like object (same as tests/golden/make_golden.py)."""
    h = 0xCBF29CE484222325
    for b in np.frombuffer(buf, np.uint8).tolist():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure), built on first use"""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def im0(golden_dir):
    """the reference's test image src/data/im0.jpg as 8-bit gray (committed fixture)"""
    from PIL import Image
    return np.asarray(Image.open(os.path.join(golden_dir, "im0_gray_1920x1080.png")))


@pytest.fixture(scope="session")
def sift():
    """the product package; GPU tests fail (not skip) when the library or device is missing"""
    # PyTorch's wheel carries its own HIP runtime.  When a test uses both, torch has to initialise
    # first: after libsift_hip.so has brought up /opt/rocm's runtime, torch.cuda fails to initialise
    # ("No HIP GPUs are available"); the other order works.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    import sift_amd
    sift_amd.load_library()
    return sift_amd
