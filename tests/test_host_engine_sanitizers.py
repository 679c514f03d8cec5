"""The host engine (ExecutionMode.CPU) under AddressSanitizer + UBSan, on the CPU: tests/native/host_engine_san.cpp drives
host_engine.cpp directly with clouds full of NaN / inf / 1e308 coordinates, non-finite values and glyph channels, footprints
that leave the grid on every side, radii of 1e9 cells and half lengths of 1e12, tiles that do not divide the grid and more
threads than rows.  Any out-of-bounds access, signed overflow or float-to-int conversion out of range aborts the harness;
and the bands must be the same bits at 1, 7 and 400 threads."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pointcloud-raster_amd")


def test_host_engine_survives_adversarial_clouds_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    host = os.path.join(PKG, "host")
    srcs = [os.path.join(ROOT, "tests", "native", "host_engine_san.cpp"), os.path.join(host, "src", "host_engine.cpp")]
    exe = str(tmp_path / "host_engine_san")
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fopenmp", "-ffp-contract=off", "-fsanitize=address,undefined,float-cast-overflow",
                    "-fno-sanitize-recover=undefined,float-cast-overflow", "-fno-omit-frame-pointer",
                    "-I", os.path.join(host, "include"), "-I", os.path.join(host, "src"), "-I", os.path.join(ROOT, "include"),
                    *srcs, "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="4")
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "host engine survived" in out.stdout
