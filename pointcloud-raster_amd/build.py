#!/usr/bin/env python3
"""Builds the in-tree native artefacts of the MI355X engine:

  csrc/*.hip           -> lib/libpcr_hip.so      (hipcc --offload-arch=gfx950; the C-ABI of include/pcr_hip.h)
  host/src/*.cpp +
  python/bindings/*.cpp -> python/pcr/_pcr*.so    (g++ -std=c++17 + pybind11; links libpcr_hip.so)

Incremental: an object is rebuilt only when its source or a header is newer.
Used by __graft_entry__.build(); `python build.py [--force] [--hip-only]`.
"""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
             "-ffp-contract=off", "-Wall", "-Wno-unused-function",
             "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]


def newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise SystemExit("build failed")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def build_hip(force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB, exist_ok=True)
    src_dir = os.path.join(HERE, "csrc")
    srcs = sorted(f for f in os.listdir(src_dir) if f.endswith(".hip"))
    hdrs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith(".hpp")]
    hdrs.append(os.path.join(ROOT, "include", "pcr_hip.h"))
    hdr_t = newest(hdrs)
    jobs, objs = [], []
    for s in srcs:
        sp = os.path.join(src_dir, s)
        op = os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_t):
            jobs.append([HIPCC] + HIP_FLAGS + ["-c", sp, "-o", op])
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    out = os.path.join(LIB, "libpcr_hip.so")
    if jobs or not os.path.exists(out):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_host(force=False):
    import pybind11
    src_dirs = [os.path.join(HERE, "host", "src"), os.path.join(HERE, "python", "bindings")]
    srcs = [os.path.join(d, f) for d in src_dirs for f in sorted(os.listdir(d)) if f.endswith(".cpp")]
    hdrs = []
    for base, _, files in os.walk(os.path.join(HERE, "host", "include")):
        hdrs += [os.path.join(base, f) for f in files]
    hdrs.append(os.path.join(ROOT, "include", "pcr_hip.h"))
    for d in src_dirs:                                   # the internal headers next to the sources
        hdrs += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")]
    hdr_t = newest(hdrs)
    cxx = os.environ.get("CXX", "g++")
    flags = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-fopenmp", "-ffp-contract=off",
             "-I" + os.path.join(HERE, "host", "include"), "-I" + os.path.join(ROOT, "include"),
             "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"]]
    jobs, objs = [], []
    for sp in srcs:
        op = os.path.join(OBJ, "host_" + os.path.basename(sp).replace(".cpp", ".o"))
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_t):
            jobs.append([cxx] + flags + ["-c", sp, "-o", op])
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(HERE, "python", "pcr", "_pcr" + ext)
    hip_so = os.path.join(LIB, "libpcr_hip.so")
    if jobs or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(hip_so):
        run([cxx, "-shared", "-o", out] + objs +
            ["-L" + LIB, "-lpcr_hip", "-lz", "-pthread", "-fopenmp", "-Wl,-rpath,$ORIGIN/../../lib", "-Wl,--no-undefined",
             "-L" + sysconfig.get_config_var("LIBDIR"), "-lpython" + sysconfig.get_config_var("LDVERSION")])
    return out


def main(argv):
    force = "--force" in argv
    so = build_hip(force)
    print("built", os.path.relpath(so, ROOT))
    if "--hip-only" not in argv:
        ext = build_host(force)
        print("built", os.path.relpath(ext, ROOT))


if __name__ == "__main__":
    main(sys.argv[1:])
