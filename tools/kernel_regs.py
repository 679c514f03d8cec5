#!/usr/bin/env python3
"""Register use of every kernel of one .hip unit, as the gfx950 code object's metadata reports it
(sgpr / vgpr counts, SPILLS, LDS, scratch).  usage: tools/kernel_regs.py csrc/scatter_binned.hip [name-substring]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pointcloud-raster_amd", "csrc"), "--cuda-device-only", "-S"]


def kernels(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        extra = os.environ.get("PCR_EXTRA_FLAGS", "").split()          # e.g. PCR_EXTRA_FLAGS="-DPCR_ROUTE_PER=16"
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    rows = []
    for block in re.split(r"\n  - ", meta)[1:]:
        f = dict(re.findall(r"\.(\w+):\s+(\S+)", block))
        if "name" not in f:
            continue
        name = subprocess.run(["c++filt", f["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::|pcrhip::", "", name).split("(")[0].replace("void ", "")
        rows.append((name, int(f.get("sgpr_count", 0)), int(f.get("sgpr_spill_count", 0)), int(f.get("vgpr_count", 0)),
                     int(f.get("vgpr_spill_count", 0)), int(f.get("group_segment_fixed_size", 0)), int(f.get("private_segment_fixed_size", 0))))
    return rows


if __name__ == "__main__":
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    print(f"{'kernel':70s} sgpr spill vgpr spill  lds scratch")
    for r in kernels(sys.argv[1]):
        if pat in r[0]:
            print(f"{r[0][:70]:70s} {r[1]:4d} {r[2]:5d} {r[3]:4d} {r[4]:5d} {r[5]:5d} {r[6]:5d}")
