"""CPU-side checks of the drop-in boundary: libpcr_hip.so loads and exports every symbol
include/pcr_hip.h declares (no compute calls: there is no GPU in the CPU suite)."""
import ctypes as C
import os
import re

from conftest import ROOT, load_cabi


def header_symbols():
    text = open(os.path.join(ROOT, "include", "pcr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcr_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    A = load_cabi()
    L = A.lib()
    names = header_symbols()
    assert len(names) >= 40
    raw = C.CDLL(A.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/pcr_hip.h but not exported"
    # and the ctypes table covers exactly the header
    assert sorted(A.SYMBOLS) == names
    assert L.pcr_hip_abi_version() == 5


def test_argument_errors_do_not_need_a_gpu():
    A = load_cabi()
    L = A.lib()
    k = C.c_int(0)
    assert L.pcr_hip_state_floats(A.AVERAGE, C.byref(k)) == 0 and k.value == 2
    assert L.pcr_hip_state_floats(A.COUNT, C.byref(k)) == 0 and k.value == 1
    assert L.pcr_hip_state_floats(7, C.byref(k)) == 1          # Percentile: not registered
    assert b"unknown reduction type" in L.pcr_hip_last_error()
    g = A.make_grid((0, 0, 4, 4))
    g.width = 0
    eng = C.c_void_p()
    assert L.pcr_hip_engine_create(C.byref(eng), C.byref(g), 0, None) == 1
    assert b"dimensions must be positive" in L.pcr_hip_last_error()
    n = C.c_int(-1)
    assert L.pcr_hip_device_count(C.byref(n)) == 0 and n.value >= 0


def test_header_is_plain_c_and_links(tmp_path):
    """include/pcr_hip.h must be consumable by a C99 compiler (cgo / JNI / ctypes all bind C), and a C program
    must link against libpcr_hip.so with nothing but the header."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "pointcloud-raster_amd", "lib")
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include "pcr_hip.h"\n'
                   'int main(void) {\n'
                   '    pcr_hip_engine* e = 0;\n'
                   '    int rc;\n'
                   '    printf("abi %d\\n", pcr_hip_abi_version());\n'
                   '    rc = pcr_hip_engine_create(&e, 0, 0, 0);      /* null grid: rejected before any HIP call */\n'
                   '    printf("rc %d msg %s\\n", rc, pcr_hip_last_error());\n'
                   '    return rc == PCR_HIP_INVALID_ARGUMENT ? 0 : 1;\n'
                   '}\n')
    gcc = shutil.which("gcc")
    assert gcc, "gcc not found"
    exe = tmp_path / "abi"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                    str(src), "-o", str(exe), "-L", lib_dir, "-lpcr_hip", f"-Wl,-rpath,{lib_dir}"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi 5" in out.stdout
