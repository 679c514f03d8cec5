"""Grids with more LDS tiles than one binning pass takes: the Point glyph sorts in two levels (groups of
tiles, then tiles), glyphs -- and the Point glyph beyond 131072 tiles -- are swept in row bands
(scatter_binned.hip / scatter_binned_glyph.hip).  PCR_HIP_DEBUG_MAX_BINS lowers the per-pass limit (8064)
so that both are reached on grids the oracle finishes in seconds; PCR_HIP_DEBUG_TWO_LEVEL=0 forces the bands.
The same bars as everywhere else apply (Count/Min/Max bit-exact, sums 1e-5, glyphs rtol 1e-4)."""
import numpy as np
import pytest

import pcr_oracle_py as O
from conftest import assert_band_close, load_cabi

pytestmark = pytest.mark.gpu
RT = {"Sum": 0, "Max": 1, "Min": 2, "Average": 3, "WeightedAverage": 4, "Count": 5}


@pytest.fixture(scope="module")
def A():
    return load_cabi()


@pytest.fixture()
def few_bins(monkeypatch):
    monkeypatch.setenv("PCR_HIP_DEBUG_MAX_BINS", "6")      # read by pcr_hip_engine_create
    yield 6


def mask_for(A, rtype):
    return {0: A.PLANE_SUM, 1: A.PLANE_MAX, 2: A.PLANE_MIN, 3: 3, 4: 3, 5: A.PLANE_WGT}[rtype]


def run_gpu(A, og, rtype, x, y, v, glyph=None, own_rows=None, halo=0, **ch):
    grid = A.make_grid((og.min_x, og.min_y, og.max_x, og.max_y), cell=(og.cell_size_x, og.cell_size_y),
                       dims=(og.width, og.height), tile=(og.tile_width, og.tile_height), own_rows=own_rows, halo=halo)
    run = A.ReductionRun(grid, mask_for(A, rtype), path=2)
    try:
        run.scatter(x, y, v, glyph=glyph, **ch)
        return run.finalize(rtype), run.stats()
    finally:
        run.close()


@pytest.mark.parametrize("mode", ["two_level", "bands"])
@pytest.mark.parametrize("rname", ["Sum", "Count", "Average", "Max", "Min"])
def test_point_large_grid_paths_match_oracle(A, few_bins, monkeypatch, rname, mode):
    monkeypatch.setenv("PCR_HIP_DEBUG_TWO_LEVEL", "1" if mode == "two_level" else "0")
    # 300 x 700 cells = 3 x 8 tiles of 128 x 96 (Sum+Count) -> at most 2 tile rows per band
    og = O.make_grid((0.0, 0.0, 300.0, 700.0), tile=(64, 64))
    rng = np.random.default_rng(11)
    n = 200_000
    x = rng.uniform(-3.0, 303.0, n)
    y = rng.uniform(-3.0, 703.0, n)
    v = rng.normal(0.0, 5.0, n).astype(np.float32)
    rt = RT[rname]
    got, st = run_gpu(A, og, rt, x, y, v)
    assert st.path == 1 and st.num_bins > few_bins, "the band sweep was not taken"
    want = O.run(og, rt, x, y, v)
    exact = O.run(og, rt, x, y, v, wide=True)
    ref = O.Reduction(og, rt)
    ref.ingest(x, y, v)
    assert st.points_valid == ref.points_valid()
    if rname in ("Count", "Max", "Min"):
        assert_band_close(got, want, what=f"{rname} bit-exact")
    else:
        assert np.array_equal(np.isnan(got), np.isnan(want))
        fin = ~np.isnan(exact)
        err = np.abs(got[fin].astype(np.float64) - exact[fin])
        tol = 1e-5 * np.maximum(1.0, np.abs(exact[fin])) * (5.0 if rname == "Sum" else 1.0)   # N(0,5) values
        assert (err <= tol).all(), f"max err {err.max()}"


@pytest.mark.parametrize("mode", ["two_level", "bands"])
@pytest.mark.parametrize("kind", ["gauss", "line"])
def test_glyph_large_grid_paths_match_oracle(A, few_bins, monkeypatch, kind, mode):
    # Gaussian index records take the two-level sort; the Line's 32-byte records always sweep bands
    monkeypatch.setenv("PCR_HIP_DEBUG_TWO_LEVEL", "1" if mode == "two_level" else "0")
    og = O.make_grid((0.0, 0.0, 200.0, 600.0), tile=(4096, 4096))
    rng = np.random.default_rng(12)
    n = 20_000
    x = rng.uniform(-2.0, 202.0, n)
    y = rng.uniform(-2.0, 602.0, n)
    v = rng.uniform(1.0, 2.0, n).astype(np.float32)
    if kind == "gauss":
        gl = dict(type=A.GLYPH_GAUSSIAN, sigma_x=2.0, sigma_y=1.5, max_radius=6.0)
        ogl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=2.0, sigma_y=1.5, max_radius=6.0)
        ch = {}
    else:
        gl = dict(type=A.GLYPH_LINE, half_length=9.0, max_radius=32.0)
        ogl = O.make_glyph(O.GLYPH_LINE, half_length=9.0, max_radius=32.0)
        ch = dict(direction=rng.uniform(0.0, 2 * np.pi, n).astype(np.float32))
    rt = RT["WeightedAverage"]
    got, st = run_gpu(A, og, rt, x, y, v, glyph=gl, **ch)
    assert st.path == 1 and st.num_bins > few_bins
    want = O.run(og, rt, x, y, v, glyph=ogl, **ch)
    exact = O.run(og, rt, x, y, v, glyph=ogl, wide=True, **ch).astype(np.float64)
    gn, wn = np.isnan(got), np.isnan(want)
    assert (gn != wn).sum() <= (0 if kind == "line" else 2)
    both = ~gn & ~wn
    err = np.abs(got[both].astype(np.float64) - exact[both])
    assert (err <= 1e-4 * np.maximum(1e-3, np.abs(exact[both]))).all()
    # Count through the same path is exact for the Line (integer weights)
    if kind == "line":
        gotc, _ = run_gpu(A, og, RT["Count"], x, y, v, glyph=gl, **ch)
        np.testing.assert_array_equal(gotc, O.run(og, RT["Count"], x, y, v, glyph=ogl, **ch))


@pytest.mark.parametrize("tile", [(4096, 4096), (64, 64)])
@pytest.mark.parametrize("sigma", [0.3, 0.6, 1.0])
def test_gauss_cell_tiles_sweep_row_bands(A, few_bins, tile, sigma):
    """The register-accumulating cell tiles (radius <= 3) under a tiny bin limit: 4 tiles across, one tile row of 20 cell
    rows per band, 30 bands - every footprint that straddles a band edge is painted by the band that owns its centre."""
    og = O.make_grid((0.0, 0.0, 200.0, 600.0), tile=tile)
    rng = np.random.default_rng(int(sigma * 10))
    n = 60_000
    x = rng.uniform(-2.0, 202.0, n)
    y = rng.uniform(-2.0, 602.0, n)
    v = rng.uniform(1.0, 2.0, n).astype(np.float32)
    gl = dict(type=A.GLYPH_GAUSSIAN, sigma_x=sigma, sigma_y=sigma, max_radius=3.0)
    ogl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=sigma, sigma_y=sigma, max_radius=3.0)
    rt = RT["WeightedAverage"]
    got, st = run_gpu(A, og, rt, x, y, v, glyph=gl)
    assert st.path == 1 and st.num_bins > 4 * few_bins, "the band sweep was not taken"
    want = O.run(og, rt, x, y, v, glyph=ogl)
    exact = O.run(og, rt, x, y, v, glyph=ogl, wide=True).astype(np.float64)
    gn, wn = np.isnan(got), np.isnan(want)
    assert (gn != wn).sum() <= 2
    both = ~gn & ~wn
    err = np.abs(got[both].astype(np.float64) - exact[both])
    assert (err <= 1e-4 * np.maximum(1e-3, np.abs(exact[both]))).all()


@pytest.mark.parametrize("mode", ["two_level", "bands"])
def test_point_large_grid_paths_inside_a_row_block_shard(A, few_bins, monkeypatch, mode):
    """Both compose with the owned-row window: a shard owning rows [200, 520) of 700."""
    monkeypatch.setenv("PCR_HIP_DEBUG_TWO_LEVEL", "1" if mode == "two_level" else "0")
    og = O.make_grid((0.0, 0.0, 300.0, 700.0))
    rng = np.random.default_rng(13)
    n = 100_000
    x, y = rng.uniform(0, 300, n), rng.uniform(0, 700, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    got, st = run_gpu(A, og, RT["Count"], x, y, v, own_rows=(200, 520))
    want = O.run(og, RT["Count"], x, y, v)[200:520]
    # untouched-tile NaN semantics are per reference tile (one 4096^2 tile here): compare counts where defined
    np.testing.assert_array_equal(np.nan_to_num(got), np.nan_to_num(want))
    assert st.points_valid == int(np.nansum(want))


def test_two_level_sort_many_tiles_and_hot_tiles(A, monkeypatch):
    """No debug limit: a 1200 x 9000 grid has 710 tiles -- fewer than 8064, so lower the limit to 100 and
    let the two-level sort run with ~27 groups; a clustered cloud puts > 16384 records into single groups and tiles."""
    monkeypatch.setenv("PCR_HIP_DEBUG_MAX_BINS", "100")
    og = O.make_grid((0.0, 0.0, 1200.0, 9000.0))
    rng = np.random.default_rng(14)
    n = 600_000
    x = np.concatenate([rng.uniform(0, 1200, n // 2), rng.normal(600.0, 15.0, n // 2)])
    y = np.concatenate([rng.uniform(0, 9000, n // 2), rng.normal(4500.0, 20.0, n // 2)])
    v = rng.normal(0.0, 1.0, n).astype(np.float32)
    for rname in ("Count", "Max"):
        got, st = run_gpu(A, og, RT[rname], x, y, v)
        assert st.path == 1 and st.num_bins > 100           # e.g. 10 x 71 tiles of 128 x 128 for one 4-byte plane
        ref = O.Reduction(og, RT[rname])
        ref.ingest(x, y, v)
        assert st.points_valid == ref.points_valid()
        assert_band_close(got, ref.finalize(), what=f"{rname} bit-exact")
