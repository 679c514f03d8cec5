"""Measured error of the moment + convolution path against the CPU oracle accumulated in double
(the number DESIGN.md section 3 quotes).  Lives under tests/ because it uses the oracle (test infrastructure).
Run on the GPU box: python tests/measure_moment_error.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))  # conftest helpers
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pcr_oracle_py as O
from conftest import load_cabi

A = load_cabi()
for sigma, maxr, G in ((16.0, 48.0, (320, 256)), (8.0, 24.0, (256, 192)), (4.0, 12.0, (256, 192))):
    W, H = G
    og = O.make_grid((0.0, 0.0, float(W), float(H)))
    rng = np.random.default_rng(5)
    n = 20000
    x, y = rng.uniform(0, W, n), rng.uniform(0, H, n)
    v = rng.normal(10.0, 3.0, n).astype(np.float32)
    gl = dict(type=A.GLYPH_GAUSSIAN, sigma_x=sigma, sigma_y=sigma, max_radius=maxr)
    ogl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=sigma, sigma_y=sigma, max_radius=maxr)
    for rname, rt, mask in (("Sum", 0, 1), ("Count", 5, 2), ("WeightedAverage", 4, 3)):
        out = {}
        for path in (2, 3):
            grid = A.make_grid((0.0, 0.0, float(W), float(H)))
            run = A.ReductionRun(grid, mask, path=path)
            run.scatter(x, y, v, glyph=gl)
            out[path] = run.finalize(rt).astype(np.float64)
            order = run.stats().lds_apron
            run.close()
        exact = O.run(og, rt, x, y, v, glyph=ogl, wide=True).astype(np.float64)
        ref32 = O.run(og, rt, x, y, v, glyph=ogl).astype(np.float64)
        m = ~np.isnan(exact)
        rel = lambda a: np.max(np.abs(a[m] - exact[m]) / np.maximum(1e-3, np.abs(exact[m])))
        print(f"sigma={sigma:g} r={int(maxr)} {rname:16s} K={order}: moments {rel(out[3]):.2e}  LDS splat {rel(out[2]):.2e}  "
              f"fp32 CPU oracle {rel(ref32):.2e}   (max rel err vs double)")
