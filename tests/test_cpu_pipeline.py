"""oracle/pcr_cpu_pipeline.cpp (the reference's CPU stages, timed by bench.py's cpu_baseline leg) must return what
the oracle returns: it only adds the reference's control flow -- assign, serial sort, per-tile batches, per-update
omp critical (src/engine/tile_router.cpp:84-240, src/ops/reduction_registry.cpp:63-92) -- around the oracle's
arithmetic.  Count / Min / Max bit-exact whatever the thread count; sums within fp32 re-association."""
import numpy as np
import pytest

import pcr_oracle_py as O
from conftest import assert_band_close


def _cloud(n, w, h, seed, margin=3.0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-margin, w + margin, n)           # some points outside the grid: sorted to the end, dropped
    y = rng.uniform(-margin, h + margin, n)
    v = rng.normal(0, 1, n).astype(np.float32)
    return x, y, v


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("rtype", [O.SUM, O.MAX, O.MIN, O.AVERAGE, O.WEIGHTED_AVERAGE, O.COUNT])
def test_point_ops_match_oracle(rtype, threads):
    x, y, v = _cloud(40000, 150, 110, seed=rtype)
    g = O.make_grid((0, 0, 150, 110), tile=(64, 48))      # 3 x 3 tiles, ragged edges
    want = O.run(g, rtype, x, y, v)
    got, stages = O.cpu_pipeline_run(g, rtype, x, y, v, threads=threads)
    exact = rtype in (O.MAX, O.MIN, O.COUNT)
    assert_band_close(got, want, rtol=0.0 if exact else 1e-5, atol=0.0 if exact else 1e-5, what=O.RTYPE_NAMES[rtype])
    assert set(stages) == {"assign", "sort", "extract_batches", "accumulate", "finalize"}


def test_untouched_tiles_stay_nan_and_empty_cloud():
    g = O.make_grid((0, 0, 128, 128), tile=(64, 64))
    x = np.array([10.5, 11.5]); y = np.array([120.0, 119.0]); v = np.array([1.0, 2.0], dtype=np.float32)
    got, _ = O.cpu_pipeline_run(g, O.SUM, x, y, v, threads=2)
    assert_band_close(got, O.run(g, O.SUM, x, y, v), what="one touched tile")
    assert np.isnan(got[:, 64:]).all() and np.isnan(got[64:, :]).all()
    e = np.zeros(0)
    got, _ = O.cpu_pipeline_run(g, O.COUNT, e, e, e.astype(np.float32))
    assert np.isnan(got).all()


def test_glyphs_match_oracle():
    x, y, v = _cloud(3000, 96, 96, seed=9, margin=0.0)
    g = O.make_grid((0, 0, 96, 96), tile=(48, 48))
    rng = np.random.default_rng(10)
    gl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=1.5, sigma_y=1.5, max_radius=5.0)
    sx = rng.uniform(0.5, 2.5, len(x)).astype(np.float32)
    got, _ = O.cpu_pipeline_run(g, O.WEIGHTED_AVERAGE, x, y, v, glyph=gl, threads=2, sigma_x=sx)
    assert_band_close(got, O.run(g, O.WEIGHTED_AVERAGE, x, y, v, glyph=gl, sigma_x=sx), rtol=1e-5, atol=1e-6, what="gaussian")
    gl = O.make_glyph(O.GLYPH_LINE, half_length=6.0, max_radius=8.0)
    d = rng.uniform(0, np.pi, len(x)).astype(np.float32)
    got, _ = O.cpu_pipeline_run(g, O.COUNT, x, y, v, glyph=gl, threads=1, direction=d)
    assert_band_close(got, O.run(g, O.COUNT, x, y, v, glyph=gl, direction=d), what="line count")
