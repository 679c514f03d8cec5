#!/usr/bin/env python3
"""Runs the REFERENCE's own CPU code (oracle/_ref/libpcr_ref.so, built by `make -C oracle ref`
from the sources under /root/reference) on the seeded cases of cases.py and stores its
outputs as fixtures: tests/golden/ref_vectors.npz.

Only runnable in the development container (needs oracle/_ref).  The fixture holds
DATA only (raw tile state and finalized tile per case); inputs are regenerated from seeds.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)

import pcr_oracle_py as O   # noqa: E402
import cases                 # noqa: E402


def run_glyph(lib_fn, init_fn, final_fn, case):
    x, y, v, ch = cases.glyph_inputs(case)
    g, t = case["grid"], case["tile"]
    grid = O.make_grid(g["bounds"], cell=g["cell"], dims=g["dims"])
    sig = case.get("sigma", (1.0, 1.0))
    gl = O.make_glyph(case["glyph"], direction=case.get("direction", 0.0),
                      half_length=case.get("half_length", 1.0), sigma_x=sig[0], sigma_y=sig[1],
                      rotation=case.get("rotation", 0.0), max_radius=case["max_radius"])
    pts, keep = O.make_points(x, y, v, **ch)
    cells = t["tw"] * t["th"]
    k = 2 if case["rtype"] in (cases.AVERAGE, cases.WEIGHTED_AVERAGE) else 1
    state = np.zeros(k * cells, dtype=np.float32)
    assert init_fn(case["rtype"], state.ctypes.data, cells) == 0
    rc = lib_fn(C.byref(gl), case["rtype"], C.byref(pts), state.ctypes.data, cells, C.byref(grid),
                t["col0"], t["row0"], t["tw"], t["th"])
    assert rc == 0, rc
    out = np.zeros(cells, dtype=np.float32)
    assert final_fn(case["rtype"], state.ctypes.data, out.ctypes.data, cells) == 0
    del keep
    return state, out


def run_point(accum_fn, init_fn, final_fn, case):
    ci, v = cases.point_inputs(case)
    cells = case["tile_cells"]
    k = 2 if case["rtype"] in (cases.AVERAGE, cases.WEIGHTED_AVERAGE) else 1
    state = np.zeros(k * cells, dtype=np.float32)
    assert init_fn(case["rtype"], state.ctypes.data, cells) == 0
    assert accum_fn(case["rtype"], ci.ctypes.data, v.ctypes.data, state.ctypes.data, len(ci), cells) == 0
    out = np.zeros(cells, dtype=np.float32)
    assert final_fn(case["rtype"], state.ctypes.data, out.ctypes.data, cells) == 0
    return state, out


def main():
    R = O.ref_lib()
    if R is None:
        sys.exit("oracle/_ref/libpcr_ref.so not built (run: make -C oracle ref)")
    out = {}
    for case in cases.GLYPH_CASES:
        st, fin = run_glyph(R.pcr_ref_accumulate_glyph, R.pcr_ref_init_state, R.pcr_ref_finalize_state, case)
        out[case["name"] + "/state"] = st
        out[case["name"] + "/final"] = fin
    for case in cases.POINT_CASES:
        st, fin = run_point(R.pcr_ref_accumulate, R.pcr_ref_init_state, R.pcr_ref_finalize_state, case)
        out[case["name"] + "/state"] = st
        out[case["name"] + "/final"] = fin
    path = os.path.join(HERE, "ref_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
