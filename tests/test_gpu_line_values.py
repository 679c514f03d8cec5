"""Adversarial VALUE patterns for the Line tiles' packed LDS window (visits << 48 | fixed-point sum, an eight-exponent window
set from the item's first 1024 values, everything else listed and walked as doubles): Count bit-exact, the sum within a few
f32 ulp of the cell's own sum of |v| against the double-accumulated oracle, NaN / inf where the reference puts them
(src/engine/glyph_kernels.cu:252-281 adds the value to every visited cell)."""
import numpy as np
import pytest

import pcr_oracle_py as O
from conftest import load_cabi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    return load_cabi()

W, H, N = 400, 300, 120000


def _values(kind):
    rng = np.random.default_rng(0)
    n = N
    if kind == "denormals_only":
        return rng.uniform(1e-45, 1e-39, n)
    if kind == "zeros_first":
        v = rng.uniform(0.5, 1.0, n)
        v[:5000] = 0.0                                     # the first 1024 values of most items say nothing about the range
        return v
    if kind == "late_large":
        v = rng.uniform(1e-3, 2e-3, n)
        v[3000::7] = 1e6                                   # 29 exponents above the window the first values set: listed
        return v
    if kind == "cancelling":
        return np.where(np.arange(n) % 2 == 0, 12345.678, -12345.678)
    if kind == "nonfinite":
        v = rng.normal(0, 1, n)
        v[::9973], v[5::9973], v[11::7919] = np.inf, -np.inf, np.nan
        return v
    if kind == "negative":
        return -rng.uniform(0, 1, n)
    if kind == "exponents_120":
        return rng.uniform(0, 1, n) * 2.0 ** rng.integers(-60, 60, n)
    if kind == "near_float_max":
        return np.full(n, 3.4e38)                          # sums overflow to inf exactly where the reference's do
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["denormals_only", "zeros_first", "late_large", "cancelling", "nonfinite", "negative",
                                  "exponents_120", "near_float_max"])
def test_line_window_value_patterns(A, kind):
    rng = np.random.default_rng(1)
    x, y = rng.uniform(0, W, N), rng.uniform(0, H, N)
    d = rng.uniform(0, np.pi, N).astype(np.float32)
    v = np.asarray(_values(kind), dtype=np.float32)
    gl = dict(type=A.GLYPH_LINE, half_length=6.0, max_radius=8.0)
    ogl = O.make_glyph(O.GLYPH_LINE, half_length=6.0, max_radius=8.0)
    og = O.make_grid((0.0, 0.0, float(W), float(H)))
    grid = A.make_grid((0.0, 0.0, float(W), float(H)), dims=(W, H))
    run = A.ReductionRun(grid, 3, path=2)
    try:
        run.scatter(x, y, v, glyph=gl, direction=d)
        got_s, got_c = run.plane("d_sum").astype(np.float64), run.plane("d_wgt")
        assert run.stats().path == 1
    finally:
        run.close()
    want_c = np.nan_to_num(O.run(og, O.COUNT, x, y, v, glyph=ogl, direction=d))
    assert np.array_equal(got_c, want_c)
    with np.errstate(all="ignore"):
        exact = O.run(og, O.SUM, x, y, v, glyph=ogl, direction=d, wide=True).astype(np.float64)
        mag = np.nan_to_num(O.run(og, O.SUM, x, y, np.abs(np.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0)), glyph=ogl,
                                  direction=d, wide=True).astype(np.float64))
    assert np.array_equal(np.isnan(exact), np.isnan(got_s))
    inf = np.isinf(exact)
    assert np.array_equal(exact[inf], got_s[inf])
    occ = np.isfinite(exact) & (want_c > 0)
    err = np.abs(got_s[occ] - exact[occ])
    assert (err <= 4e-7 * np.maximum(mag[occ], 1e-45) + 1.5e-45).all(), \
        f"{kind}: max rel err {np.max(err / np.maximum(mag[occ], 1e-45)):.3e}"
