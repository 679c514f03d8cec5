"""Known-answer PCRP file built byte by byte from the format specification in the reference header
(include/pcr/io/point_cloud_io.h:22-39) -- NOT with any reader/writer of this repository, and not
with the reference's own writer (src/io/point_cloud_io.cpp links against src/core/types.cpp, which
needs <proj.h>: unbuildable in this image).

    python tests/golden/make_pcrp_fixture.py      -> tests/golden/spec_cloud.pcrp

Layout: u32 "PCRP" | u32 version=1 | u64 n | u32 nch | u32 wkt_len | wkt |
        {u16 name_len, name, u8 dtype} x nch | f64 x[n] | f64 y[n] | channel arrays in table order.
DataType bytes (include/pcr/core/types.h): Float32=0, Float64=1, Int32=2, UInt32=3."""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
WKT = 'PROJCS["fixture"]'


def cloud():
    n = 7
    x = np.array([0.0, 1.5, -2.25, 1e6 + 0.125, np.inf, -np.inf, np.nan])
    y = np.arange(n, dtype=np.float64) * 0.5 - 1.0
    chans = [  # table order = lexicographic (the order this build's writer uses)
        ("classification", 2, (np.arange(n) - 3).astype(np.int32)),
        ("intensity", 0, np.array([0.1, 2.0, np.inf, -np.inf, np.nan, -0.0, 3.5], dtype=np.float32)),
        ("time", 1, np.linspace(0.0, 1.0, n)),
    ]
    return n, x, y, chans


def build():
    n, x, y, chans = cloud()
    b = struct.pack("<IIQI", 0x50524350, 1, n, len(chans))
    b += struct.pack("<I", len(WKT)) + WKT.encode()
    for name, dt, _ in chans:
        b += struct.pack("<H", len(name)) + name.encode() + struct.pack("<B", dt)
    b += x.astype("<f8").tobytes() + y.astype("<f8").tobytes()
    for _, _, a in chans:
        b += a.tobytes()
    return b


if __name__ == "__main__":
    with open(os.path.join(HERE, "spec_cloud.pcrp"), "wb") as f:
        f.write(build())
    print("wrote spec_cloud.pcrp,", len(build()), "bytes")
