"""csrc/libm_sincosf.hpp -- glibc's sinf / cosf restated for the device -- against the system's libm, bit for bit, on the CPU
(tests/native/sincosf_check.cpp: ~3.3e7 arguments: every fifth float of [pi/4, 2 pi), the neighbourhoods of 79 multiples of
pi/2, the seam at 120, 1.2e7 random bit patterns, specials).  The reference's CPU path takes std::cos / std::sin of a float
(src/engine/glyph_kernels.cu:126-128, 236-237) and ROUNDS a Line's end points to cells (:245-250): the device needs the same
last bit, and (float)cos((double)a) is not it (2.7 % of random arguments differ: glibc's float routines carry up to 0.56 ulp)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_sincosf_is_glibcs_bit_for_bit(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "sincosf_check")
    flags = ["-std=c++17", "-O2", "-ffp-contract=off"]
    if "fma" in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split():
        flags.append("-mfma")                       # (without it __builtin_fma is libm's software fma: exact as well, only slow)
    subprocess.run([gxx, *flags, "-I", os.path.join(ROOT, "pointcloud-raster_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "sincosf_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 differ" in out.stdout


def test_correct_rounding_is_not_what_glibc_returns():
    """Why the restatement exists: (float)cos((double)a), the correctly rounded value rounds 1-4 used, differs from glibc's cosf
    for a few per cent of random directions."""
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.cosf.restype, libm.cosf.argtypes = C.c_float, [C.c_float]
    rng = np.random.default_rng(0)
    d = rng.uniform(0, 6.3, 20000).astype(np.float32)
    glibc = np.array([libm.cosf(float(v)) for v in d], dtype=np.float32)
    rounded = np.cos(d.astype(np.float64)).astype(np.float32)
    differ = (glibc.view(np.uint32) != rounded.view(np.uint32)).mean()
    assert 0.005 < differ < 0.08, differ
