"""BASELINE.json configs that had no GPU test in round 1:

  C4  clustered cloud (10 000 hotspots, N(0, 2 cells)), Point Max / Min (+ Count): at oracle scale bit-exact on every
      scatter path; at 50 M / 4096^2 through properties (binned == direct bit for bit, counts conserved).
  C3  50 M / 4096^2: Line hl=16 with per-point direction (binned 32-byte records vs direct atomics: Count bit-exact),
      Gaussian sigma=4 r<=12 (moments vs LDS splat vs direct within 1e-4, total weight conserved).
  C5  the shard a rank holds at N = 8: a 16384 x 2048 row window of the 16384^2 grid, Gaussian sigma=1 r<=4 -- the
      two-level sort against direct atomics on 125 M points, and the oracle on the sub-rectangle a 200 k-point cluster
      touches (no debug knobs).
The recipes are bench.py's (make_points): the same clouds the driver times."""
import numpy as np
import pytest

import pcr
import pcr_oracle_py as O
from conftest import assert_band_close
from test_gpu_pipeline_api import cloud_from, config_for, spec

pytestmark = pytest.mark.gpu


def clustered(n, G, seed, k=10_000, sigma=2.0):
    rng = np.random.default_rng(seed)                    # bench.py make_points("C4", ...)
    cx, cy = rng.uniform(2, G - 2, k), rng.uniform(2, G - 2, k)
    idx = np.arange(n) % k
    x = np.clip(cx[idx] + rng.normal(0, sigma, n), 0, G)
    y = np.clip(cy[idx] + rng.normal(0, sigma, n), 0, G)
    v = rng.uniform(0, 1, n).astype(np.float32)
    return x, y, v


def bands(p):
    return [np.array(p.result().band_array(i)) for i in range(p.result().num_bands())]


@pytest.mark.parametrize("path", [0, 1, 2], ids=["auto", "direct", "binned"])
def test_c4_clustered_max_min_count_bit_exact_vs_oracle(path):
    G, n = 1024, 2_000_000
    x, y, v = clustered(n, G, seed=42)
    og = O.make_grid((0, 0, G, G))
    p = pcr.Pipeline.create(config_for(og, [spec("Max"), spec("Min"), spec("Count")], scatter_path=path))
    p.ingest(cloud_from(x, y, {"value": v}, "device"))
    p.finalize()
    mx, mn, ct = bands(p)
    assert_band_close(mx, O.run(og, O.MAX, x, y, v), what="C4 max")
    assert_band_close(mn, O.run(og, O.MIN, x, y, v), what="C4 min")
    assert_band_close(ct, O.run(og, O.COUNT, x, y, v), what="C4 count")
    if path:
        assert p.last_scatter()["path"] == ("direct" if path == 1 else "binned")


def test_c4_full_size_paths_agree_bit_for_bit():
    G, n = 4096, 50_000_000
    x, y, v = clustered(n, G, seed=42)
    og = O.make_grid((0, 0, G, G))
    cloud = cloud_from(x, y, {"value": v}, "device")
    out = {}
    for path in (2, 1):
        p = pcr.Pipeline.create(config_for(og, [spec("Max"), spec("Min"), spec("Count")], scatter_path=path))
        p.ingest(cloud)
        p.finalize()
        out[path] = bands(p)
        assert p.last_scatter()["points_valid"] == n          # clipped to the bounds: every point is inside
        del p
    for b, name in enumerate(("max", "min", "count")):
        assert np.array_equal(out[2][b], out[1][b], equal_nan=True), f"C4 50M: binned != direct for {name}"
    ct = out[2][2]
    assert np.nansum(ct.astype(np.float64)) == n
    occ = ~np.isnan(ct) & (ct > 0)
    assert (out[2][1][occ] <= out[2][0][occ]).all()               # min <= max wherever a point fell
    # the hotspots really are hot: atomic-contention worst case (thousands of points on a few cells)
    assert ct[occ].max() > 300


def test_c3_full_size_line_binned_vs_direct():
    G, n = 4096, 50_000_000
    rng = np.random.default_rng(42)
    x, y = rng.uniform(2, G - 2, n), rng.uniform(2, G - 2, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    d = rng.uniform(0, np.pi, n).astype(np.float32)
    og = O.make_grid((0, 0, G, G))
    cloud = cloud_from(x, y, {"value": v, "direction": d}, "device")
    ls = pcr.line_splat_spec("value", direction_channel="direction", default_half_length=16.0, max_radius_cells=18.0)
    lc = pcr.line_splat_spec("value", direction_channel="direction", default_half_length=16.0, max_radius_cells=18.0)
    ls.type, lc.type = pcr.ReductionType.Sum, pcr.ReductionType.Count
    out = {}
    for path in (2, 1):
        p = pcr.Pipeline.create(config_for(og, [ls, lc], scatter_path=path))
        p.ingest(cloud)
        p.finalize()
        out[path] = bands(p)
        assert p.last_scatter()["path"] == ("direct" if path == 1 else "binned")
        del p
    assert np.array_equal(out[2][1], out[1][1], equal_nan=True), "Line count: binned != direct"
    assert_band_close(out[2][0], out[1][0], rtol=1e-4, atol=1e-4, what="Line sum binned vs direct")
    # a segment of half length 16 paints at most 2*16+1 cells per axis: total cell visits per point in [17, 47]
    total = np.nansum(out[2][1].astype(np.float64))
    assert 17 * n < total < 47 * n
    # oracle on a small window of the same cloud: points whose segment can reach rows/cols [1000, 1200)
    lo, hi, m = 1000, 1200, 19
    sel = (x >= lo - m) & (x < hi + m) & (y >= G - hi - m) & (y < G - lo + m)
    want = O.run(og, O.COUNT, x[sel], y[sel], v[sel], glyph=O.make_glyph(O.GLYPH_LINE, half_length=16.0, max_radius=18.0),
                 direction=d[sel])
    w, g = np.nan_to_num(want[lo:hi, lo:hi]), np.nan_to_num(out[2][1][lo:hi, lo:hi])
    assert np.array_equal(w, g), "Line count differs from the oracle on the [1000, 1200)^2 window"


def test_c3_full_size_gaussian_sigma4_three_paths():
    G, n = 4096, 50_000_000
    rng = np.random.default_rng(42)
    x, y = rng.uniform(2, G - 2, n), rng.uniform(2, G - 2, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    og = O.make_grid((0, 0, G, G))
    cloud = cloud_from(x, y, {"value": v}, "device")
    gw = pcr.gaussian_splat_spec("value", default_sigma=4.0, max_radius_cells=12.0)
    gc = pcr.gaussian_splat_spec("value", default_sigma=4.0, max_radius_cells=12.0)
    gc.type = pcr.ReductionType.Count                               # the weight plane itself
    out, totals = {}, {}
    for path, name in ((3, "moments"), (2, "binned"), (1, "direct")):
        p = pcr.Pipeline.create(config_for(og, [gw, gc], scatter_path=path))
        p.ingest(cloud)
        p.finalize()
        out[path] = bands(p)
        assert p.last_scatter()["path"] == name
        totals[path] = np.nansum(out[path][1].astype(np.float64))
        del p
    for path in (3, 2):
        assert_band_close(out[path][0], out[1][0], rtol=1e-4, atol=1e-6, what=f"sigma=4 weighted average, path {path} vs direct")
        assert_band_close(out[path][1], out[1][1], rtol=1e-4, atol=1e-6, what=f"sigma=4 weight plane, path {path} vs direct")
        assert abs(totals[path] - totals[1]) <= 1e-5 * totals[1]
    # total weight = N x (mean footprint weight); the mean from the oracle on a 20 k-point sample of interior points
    sl = np.nonzero((x > 20) & (x < 200) & (y > G - 200) & (y < G - 20))[0][:20000]
    og_s = O.make_grid((0, G - 220, 220, G))
    ws = O.run(og_s, O.COUNT, x[sl], y[sl], v[sl], glyph=O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=4.0, sigma_y=4.0, max_radius=12.0))
    mean_w = np.nansum(ws.astype(np.float64)) / len(sl)
    assert abs(totals[1] / n - mean_w) < 0.01 * mean_w               # edge points lose a little: within 1 %


def test_c5_shard_window_gaussian_sigma1_two_level_vs_direct_and_oracle():
    G, R0, R1 = 16384, 2048, 4096                                     # rank 1 of 8
    n_bg, n_cl = 125_000_000 - 200_000, 200_000
    rng = np.random.default_rng(43)
    # uniform background over the shard's rows, minus a 360 x 360 hole; a 200 k-point cluster inside the hole
    hx0, hy_row0, hole, m = 5000, 2600, 360, 30
    xb = rng.uniform(2, G - 2, n_bg)
    yb = rng.uniform(G - R1, G - R0, n_bg)
    in_hole = (xb >= hx0) & (xb < hx0 + hole) & (yb > G - (hy_row0 + hole)) & (yb <= G - hy_row0)
    xb[in_hole] += hole + 8                                           # push them out (still uniform elsewhere)
    xc = rng.normal(hx0 + hole / 2, 40.0, n_cl).clip(hx0 + m, hx0 + hole - m)
    yc = rng.normal(G - (hy_row0 + hole / 2), 40.0, n_cl).clip(G - (hy_row0 + hole) + m, G - hy_row0 - m)
    x, y = np.concatenate([xb, xc]), np.concatenate([yb, yc])
    v = rng.uniform(0, 1, len(x)).astype(np.float32)
    og = O.make_grid((0, 0, G, G))
    gs = pcr.gaussian_splat_spec("value", default_sigma=1.0, max_radius_cells=4.0)
    gs.type = pcr.ReductionType.Average
    cloud = cloud_from(x, y, {"value": v}, "device")
    out = {}
    for path in (0, 1):
        p = pcr.Pipeline.create(config_for(og, [gs, spec("Count")], scatter_path=path, shard_row_begin=R0, shard_row_end=R1,
                                           gpu_pool_size_bytes=24 * len(x)))
        assert p.halo_rows() == 4 and p.state_row_begin() == R0 - 4 and p.state_row_count() == (R1 - R0) + 8
        p.ingest(cloud)
        p.finalize()
        out[path] = bands(p)
        info = p.last_scatter()
        assert info["points_valid"] == len(x)
        if path == 0:
            assert info["path"] == "binned", info                     # the production choice for this shard
        del p
    assert np.array_equal(out[0][1], out[1][1], equal_nan=True), "C5 shard: Point count differs between paths"
    assert_band_close(out[0][0], out[1][0], rtol=1e-4, atol=1e-6, what="C5 shard gaussian sigma=1: binned vs direct")
    # oracle on the hole: only cluster points can reach it (background kept >= 8 cells away, r = 4)
    sub = O.make_grid((hx0, G - (hy_row0 + hole), hx0 + hole, G - hy_row0))
    sel = slice(n_bg, None)
    want = O.run(sub, O.AVERAGE, x[sel], y[sel], v[sel], glyph=O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=1.0, sigma_y=1.0, max_radius=4.0))
    got = out[0][0][hy_row0 - R0: hy_row0 - R0 + hole, hx0: hx0 + hole]
    inner = (slice(8, hole - 8), slice(8, hole - 8))                   # pushed-out background may touch the rim columns
    w, g = want[inner], got[inner]
    both = ~np.isnan(w)
    assert np.array_equal(np.isnan(g), ~both), "C5 shard: NaN mask differs from the oracle inside the cluster window"
    assert (np.abs(g[both] - w[both]) <= 1e-6 + 1e-4 * np.abs(w[both])).all()


def test_c2_full_size_sum_count_average_vs_oracle_window():
    """BASELINE configs[1] at full size (50 M uniform points, 4096^2, Sum + Count + Average on one channel, the cloud the
    driver times): every cell of a 256^2 window against the oracle run on the points that fall into it (VERDICT r02,
    weak 2: full-size C2 was property-only)."""
    G, n = 4096, 50_000_000
    rng = np.random.default_rng(42)                                 # bench.py make_points("C2", ...)
    x, y = rng.uniform(2, G - 2, n), rng.uniform(2, G - 2, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    og = O.make_grid((0, 0, G, G))
    p = pcr.Pipeline.create(config_for(og, [spec("Sum"), spec("Count"), spec("Average")], scatter_path=0))
    p.ingest(cloud_from(x, y, {"value": v}, "device"))
    p.finalize()
    assert p.last_scatter()["path"] == "binned" and p.last_scatter()["points_valid"] == n
    sm, ct, av = bands(p)
    lo, hi = 1900, 2156                                             # a window that straddles LDS-tile edges (128 x 96 tiles)
    sel = (x >= lo) & (x < hi) & (y > G - hi) & (y <= G - lo)      # rows lo..hi-1 <=> y in (G - hi, G - lo]
    want_s = O.run(og, O.SUM, x[sel], y[sel], v[sel], wide=True)
    want_c = O.run(og, O.COUNT, x[sel], y[sel], v[sel])
    want_a = O.run(og, O.AVERAGE, x[sel], y[sel], v[sel], wide=True)
    ws, wc, wa = want_s[lo:hi, lo:hi], want_c[lo:hi, lo:hi], want_a[lo:hi, lo:hi]
    gs, gc, ga = sm[lo:hi, lo:hi], ct[lo:hi, lo:hi], av[lo:hi, lo:hi]
    assert np.array_equal(np.nan_to_num(wc), np.nan_to_num(gc)), "C2 count differs from the oracle on the window"
    occ = ~np.isnan(wc)
    assert occ.sum() > 0.9 * occ.size                               # ~3 points per cell
    assert np.array_equal(np.isnan(ga), ~occ)
    assert (np.abs(gs[occ].astype(np.float64) - ws[occ]) <= 1e-5 * np.maximum(1.0, np.abs(ws[occ]))).all()
    assert (np.abs(ga[occ].astype(np.float64) - wa[occ]) <= 1e-5 * np.maximum(1.0, np.abs(wa[occ]))).all()
    assert (gs[~occ] == 0.0).all()                                  # Q2: Sum of an empty cell inside a touched tile


@pytest.mark.parametrize("sigma,max_r,win", [(1.0, 4.0, 128), (16.0, 64.0, 48)], ids=["sigma1_cell_tiles", "sigma16_moments"])
def test_bench_gaussian_clouds_full_size_vs_oracle_window(sigma, max_r, win):
    """The two Gaussian clouds of bench.py's per_glyph legs exactly as the driver times them (50 M uniform points, 4096^2,
    seed 42; sigma = 1 through the register-accumulating cell tiles, sigma = 16 through moments + the matrix-core column pass):
    every cell of a window against the oracle run on the points whose footprint can reach it."""
    G, n = 4096, 50_000_000
    rng = np.random.default_rng(42)                                 # bench.py make_points(...)
    x, y = rng.uniform(2, G - 2, n), rng.uniform(2, G - 2, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    og = O.make_grid((0, 0, G, G))
    gs = pcr.gaussian_splat_spec("value", default_sigma=sigma, max_radius_cells=max_r)
    p = pcr.Pipeline.create(config_for(og, [gs], scatter_path=0, gpu_pool_size_bytes=24 * n + (64 << 20)))
    p.ingest(cloud_from(x, y, {"value": v}, "device"))
    p.finalize()
    info = p.last_scatter()
    assert info["points_valid"] == n and info["path"] == ("binned" if sigma == 1.0 else "moments"), info
    got = bands(p)[0]
    del p
    reach = int(np.ceil(min(3.0 * sigma, max_r)))
    lo = 2010                                                        # straddles tile edges of both forms (58 x 20, 64 x 16)
    hi = lo + win
    sel = (x >= lo - reach - 1) & (x < hi + reach + 1) & (y > G - hi - reach - 1) & (y <= G - lo + reach + 1)
    ogl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=sigma, sigma_y=sigma, max_radius=max_r)
    want = O.run(og, O.WEIGHTED_AVERAGE, x[sel], y[sel], v[sel], glyph=ogl)[lo:hi, lo:hi]
    exact = O.run(og, O.WEIGHTED_AVERAGE, x[sel], y[sel], v[sel], glyph=ogl, wide=True)[lo:hi, lo:hi].astype(np.float64)
    g = got[lo:hi, lo:hi]
    assert np.array_equal(np.isnan(g), np.isnan(want)), "NaN mask differs from the oracle on the window"
    both = ~np.isnan(want)
    assert both.sum() == both.size                                   # ~3 points per cell: every cell is reached
    err = np.abs(g[both].astype(np.float64) - exact[both])
    assert (err <= 1e-4 * np.maximum(1e-3, np.abs(exact[both]))).all(), f"max rel err {np.max(err / np.abs(exact[both])):.3g}"
