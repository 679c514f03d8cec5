#!/usr/bin/env python3
"""Writes tests/golden/ref_tile_0001_0002.pcrt with the REFERENCE's own writer
(src/io/tile_state_io.cpp via oracle/_ref/libpcr_ref.so): tile (row 1, col 2), 5 cols x 3 rows,
Average state (2 floats per cell), values state[f, y, x] = 100 f + 10 y + x + 0.5.
Development container only (needs oracle/_ref)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import pcr_oracle_py as O   # noqa: E402


def fixture_state():
    f, y, x = np.meshgrid(np.arange(2), np.arange(3), np.arange(5), indexing="ij")
    return (100 * f + 10 * y + x + 0.5).astype(np.float32)


if __name__ == "__main__":
    R = O.ref_lib()
    if R is None:
        sys.exit("oracle/_ref/libpcr_ref.so not built (run: make -C oracle ref)")
    st = fixture_state()
    path = os.path.join(HERE, "ref_tile_0001_0002.pcrt")
    R.pcr_ref_write_tile_state.argtypes = [C.c_char_p] + [C.c_int] * 6 + [C.c_void_p]
    rc = R.pcr_ref_write_tile_state(path.encode(), 1, 2, 5, 3, 2, 3, st.ctypes.data)   # 3 = ReductionType::Average
    assert rc == 0
    print("wrote", path, os.path.getsize(path), "bytes")
