"""The file parsers (PCRP / CSV / GeoTIFF) under AddressSanitizer + UBSan, on the CPU: tests/native/io_fuzz.cpp writes
valid files in every layout, checks the round trip, then feeds ~1800 randomly corrupted and truncated variants to every
reader entry point.  Any out-of-bounds access, overflow or giant allocation aborts the harness."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pointcloud-raster_amd")


def test_parsers_survive_corrupt_files_under_asan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    host = os.path.join(PKG, "host")
    srcs = [os.path.join(ROOT, "tests", "native", "io_fuzz.cpp")] + [os.path.join(host, "src", f) for f in
            ("grid_io.cpp", "point_cloud_io.cpp", "point_cloud.cpp", "grid.cpp", "core.cpp")]
    exe = str(tmp_path / "io_fuzz")
    lib = os.path.join(PKG, "lib")
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-fno-omit-frame-pointer", "-I", os.path.join(host, "include"), "-I", os.path.join(host, "src"),
                    "-I", os.path.join(ROOT, "include"), *srcs, "-L", lib, "-lpcr_hip", "-lz", "-pthread",
                    f"-Wl,-rpath,{lib}", "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "parsers survived" in out.stdout
