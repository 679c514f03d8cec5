"""Parity of the HIP path, driven through the C-ABI (include/pcr_hip.h), against the CPU oracle.

Bars:
  * Count / Min / Max: bit-exact, NaN mask exact.
  * Sum / Average / WeightedAverage, Point glyph: NaN mask exact; |gpu - exact| <= 1e-5 * max(1, |exact|)
    where `exact` is the oracle accumulated in double (fp32 atomics re-associate the sum; the
    reference itself only promises 1e-5 between thread counts, tests/cpp/test_threading.cpp:97-144).
  * Gaussian / Line: NaN mask exact, values rtol 1e-4 (the reference's own CPU<->GPU criterion,
    scripts/patterns/compare_cpu_gpu_patterns.py:28,92).
"""
import os
import sys

import numpy as np
import pytest

import pcr_oracle_py as O
from conftest import assert_band_close, grid_from_json, load_cabi

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import cases  # noqa: E402

pytestmark = pytest.mark.gpu

RT = {"Sum": 0, "Max": 1, "Min": 2, "Average": 3, "WeightedAverage": 4, "Count": 5}
PATHS = [1, 2, 0]     # glyphs: direct, forced binned LDS tiles, auto
POINT_PATHS = [1, 2, 0]  # Point glyph: direct, forced binned LDS tiles, auto


def mask_for(A, rtype):
    return {0: A.PLANE_SUM, 1: A.PLANE_MAX, 2: A.PLANE_MIN, 3: A.PLANE_SUM | A.PLANE_WGT,
            4: A.PLANE_SUM | A.PLANE_WGT, 5: A.PLANE_WGT}[rtype]


def cabi_grid(A, og, own_rows=None, halo=0):
    return A.make_grid((og.min_x, og.min_y, og.max_x, og.max_y), cell=(og.cell_size_x, og.cell_size_y),
                       dims=(og.width, og.height), tile=(og.tile_width, og.tile_height),
                       own_rows=own_rows, halo=halo)


def gpu_run(A, og, rtype, clouds, glyph=None, path=0, mask=None):
    run = A.ReductionRun(cabi_grid(A, og), mask if mask is not None else mask_for(A, rtype), path=path)
    try:
        for cl in clouds:
            ch = {k: v for k, v in cl.items() if k not in ("x", "y", "value")}
            run.scatter(cl["x"], cl["y"], cl["value"], glyph=glyph, **ch)
        return run.finalize(rtype), run.stats()
    finally:
        run.close()


@pytest.fixture(scope="module")
def A():
    mod = load_cabi()
    assert mod.device_count() >= 1, "no HIP device visible"
    return mod


@pytest.mark.parametrize("path", POINT_PATHS)
def test_reference_pipeline_known_answers(A, known_answers, denan, path):
    for case in known_answers["pipeline"]:
        og = grid_from_json(O, case["grid"])
        for b, red in enumerate(case["reductions"]):
            got, _ = gpu_run(A, og, RT[red["type"]], case["clouds"], path=path)
            want = np.array(denan(case["expected"][b]), dtype=np.float32).reshape(og.height, og.width)
            assert_band_close(got, want, what=f'{case["name"]}[{red["type"]}] path={path}')


def uniform_cloud(n, g, seed, margin=2.0, value="uniform"):
    rng = np.random.default_rng(seed)
    x = rng.uniform(g.min_x + margin, g.max_x - margin, n)
    y = rng.uniform(g.min_y + margin, g.max_y - margin, n)
    if value == "uniform":
        v = rng.uniform(0.0, 1.0, n).astype(np.float32)
    else:
        v = rng.normal(0.0, 50.0, n).astype(np.float32)
    return x, y, v


@pytest.mark.parametrize("path", POINT_PATHS)
@pytest.mark.parametrize("rname", ["Sum", "Max", "Min", "Average", "WeightedAverage", "Count"])
@pytest.mark.parametrize("gridspec", [
    dict(bounds=(0, 0, 256, 256), tile=(4096, 4096)),          # one tile
    dict(bounds=(0, 0, 300, 200), tile=(64, 48)),              # ragged multi-tile, W % 4 == 0
    dict(bounds=(10, -20, 137, 81), tile=(50, 50)),            # odd width (scalar finalize), offset origin
], ids=["256sq", "300x200_tiles", "127x101_odd"])
def test_point_ops_random(A, rname, gridspec, path):
    og = O.make_grid(gridspec["bounds"], tile=gridspec["tile"])
    x, y, v = uniform_cloud(60000, og, seed=42, margin=-3.0, value="normal")   # some points out of bounds
    # leave tile (0,0) of the multi-tile grids empty to exercise untouched tiles (Q3)
    if og.tile_width < og.width:
        keep = ~((x < og.min_x + og.tile_width * og.cell_size_x) & (y > og.max_y + og.tile_height * og.cell_size_y))
        x, y, v = x[keep], y[keep], v[keep]
    rt = RT[rname]
    got, st = gpu_run(A, og, rt, [dict(x=x, y=y, value=v)], path=path)
    ref = O.Reduction(og, rt)
    ref.ingest(x, y, v)
    want = ref.finalize()
    assert st.points_in == len(x) and st.points_valid == ref.points_valid()
    if rname in ("Max", "Min", "Count"):
        assert_band_close(got, want, what=f"{rname} bit-exact")
    else:
        exact = O.run(og, rt, x, y, v, wide=True)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        fin = ~np.isnan(exact)
        err = np.abs(got[fin].astype(np.float64) - exact[fin])
        tol = 1e-5 * np.maximum(1.0, np.abs(exact[fin])) * (50.0 if rname == "Sum" else 1.0)
        # Sum of N(0,50) values: scale the absolute floor by the value scale
        assert (err <= tol).all(), f"max err {err.max()} (tol {tol.min()})"
        assert_band_close(got, want, rtol=2e-5, atol=2e-3, what=f"{rname} vs fp32 oracle")


@pytest.mark.parametrize("path", POINT_PATHS)
def test_fused_planes_one_pass(A, path):
    """Sum + Count + Average + Max + Min from ONE scatter over the points (plane_mask = 15)."""
    og = O.make_grid((0, 0, 200, 120), tile=(4096, 4096))
    x, y, v = uniform_cloud(40000, og, seed=7)
    run = A.ReductionRun(cabi_grid(A, og), 15, path=path)
    try:
        run.scatter(x, y, v)
        for rname in ("Sum", "Count", "Average", "Max", "Min"):
            got = run.finalize(RT[rname])
            want = O.run(og, RT[rname], x, y, v)
            if rname in ("Max", "Min", "Count"):
                assert_band_close(got, want, what=rname)
            else:
                assert_band_close(got, want, rtol=1e-5, atol=1e-6, what=rname)
    finally:
        run.close()


@pytest.mark.parametrize("path", POINT_PATHS)
def test_bounds_edges_q1(A, path):
    # inclusive bounds + clamp (Q1): corners and edges land in the outermost cells
    og = O.make_grid((0, 0, 4, 4))
    x = np.array([0.0, 4.0, 4.0, 2.0, 0.0, -1e-9, 4.0 + 1e-9, np.nan, 2.0])
    y = np.array([0.0, 4.0, 0.0, 4.0, 4.0, 2.0, 2.0, 2.0, np.nan])
    v = np.ones(9, dtype=np.float32)
    got, st = gpu_run(A, og, RT["Count"], [dict(x=x, y=y, value=v)], path=path)
    want = O.run(og, RT["Count"], x, y, v)
    assert_band_close(got, want, what="Q1")
    assert st.points_valid == 5
    assert got[3, 0] == 1 and got[0, 3] == 1 and got[3, 3] == 1 and got[0, 2] == 1 and got[0, 0] == 1


@pytest.mark.parametrize("path", POINT_PATHS)
def test_special_values_min_max_sum(A, path):
    og = O.make_grid((0, 0, 4, 1))
    x = np.array([0.5, 0.5, 1.5, 1.5, 2.5, 2.5, 3.5])
    y = np.full(7, 0.5)
    v = np.array([np.nan, 3.0, -0.0, -5.0, np.inf, 1.0, -np.inf], dtype=np.float32)
    for rname in ("Max", "Min", "Sum", "Count"):
        got, _ = gpu_run(A, og, RT[rname], [dict(x=x, y=y, value=v)], path=path)
        want = O.run(og, RT[rname], x, y, v)
        assert_band_close(got, want, what=rname)


@pytest.mark.parametrize("path", POINT_PATHS)
def test_empty_and_all_out_of_bounds(A, path):
    og = O.make_grid((0, 0, 16, 16), tile=(8, 8))
    e = np.zeros(0)
    got, st = gpu_run(A, og, RT["Sum"], [dict(x=e, y=e, value=e.astype(np.float32))], path=path)
    assert np.isnan(got).all() and st.points_valid == 0
    x = np.array([-5.0, 100.0]); y = np.array([3.0, 3.0]); v = np.ones(2, dtype=np.float32)
    got, st = gpu_run(A, og, RT["Sum"], [dict(x=x, y=y, value=v)], path=path)
    assert np.isnan(got).all() and st.points_valid == 0


@pytest.mark.parametrize("path", POINT_PATHS)
def test_state_survives_finalize_and_multi_ingest(A, path):
    # quirk Q9: finalize does not reset; a second ingest accumulates on top
    og = O.make_grid((0, 0, 64, 64))
    x1, y1, v1 = uniform_cloud(5000, og, 1)
    x2, y2, v2 = uniform_cloud(7000, og, 2)
    run = A.ReductionRun(cabi_grid(A, og), A.PLANE_SUM | A.PLANE_WGT, path=path)
    ref = O.Reduction(og, RT["Average"])
    try:
        run.scatter(x1, y1, v1); ref.ingest(x1, y1, v1)
        assert_band_close(run.finalize(RT["Average"]), ref.finalize(), rtol=1e-5, atol=1e-6, what="first")
        run.scatter(x2, y2, v2); ref.ingest(x2, y2, v2)
        assert_band_close(run.finalize(RT["Average"]), ref.finalize(), rtol=1e-5, atol=1e-6, what="second")
    finally:
        run.close()


def glyph_dict(A, case):
    sig = case.get("sigma", (1.0, 1.0))
    return dict(type=case["glyph"], direction=case.get("direction", 0.0),
                half_length=case.get("half_length", 1.0), sigma_x=sig[0], sigma_y=sig[1],
                rotation=case.get("rotation", 0.0), max_radius=case["max_radius"])


def oracle_glyph(case):
    sig = case.get("sigma", (1.0, 1.0))
    return O.make_glyph(case["glyph"], direction=case.get("direction", 0.0),
                        half_length=case.get("half_length", 1.0), sigma_x=sig[0], sigma_y=sig[1],
                        rotation=case.get("rotation", 0.0), max_radius=case["max_radius"])


def assert_glyph_close(got, want, exact, what, is_line):
    gn, wn = np.isnan(got), np.isnan(want)
    if is_line:
        assert np.array_equal(gn, wn), f"{what}: NaN mask"
    else:
        # a cell whose only contributions sit within an ulp of the 1e-6 cut-off may flip
        diff = gn != wn
        assert diff.sum() <= max(2, int(1e-4 * gn.size)), f"{what}: NaN mask differs in {int(diff.sum())} cells"
    both = ~gn & ~wn
    err = np.abs(got[both].astype(np.float64) - exact[both])
    tol = 1e-4 * np.maximum(1e-3, np.abs(exact[both]))
    assert (err <= tol).all(), f"{what}: max rel err {np.max(err / np.maximum(1e-3, np.abs(exact[both])))}"


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("case", cases.GLYPH_CASES, ids=lambda c: c["name"])
def test_glyph_cases_full_grid(A, case, path):
    """The seeded glyph cases of tests/golden/cases.py, through the whole path on a grid whose
    reference tile equals the case's tile rectangle (so tile clipping is live)."""
    g, t = case["grid"], case["tile"]
    # tiles of (tw, th) only tile the grid exactly when the rect is at a multiple: use the rect size
    # as tile size only for the full-tile cases; sub-rect cases use tile = rect dims with origin at rect
    if (t["col0"], t["row0"]) != (0, 0):
        tile = (16, 8)     # rect 32..80 x 16..56 is a union of 16x8 tiles -> clip per 16x8 tile
    else:
        tile = (t["tw"], t["th"])
    og = O.make_grid(g["bounds"], cell=g["cell"], tile=tile, dims=g["dims"])
    x, y, v, ch = cases.glyph_inputs(case)
    gl = glyph_dict(A, case)
    got, st = gpu_run(A, og, case["rtype"], [dict(x=x, y=y, value=v, **ch)], glyph=gl, path=path)
    ogl = oracle_glyph(case)
    want = O.run(og, case["rtype"], x, y, v, glyph=ogl, **ch)
    exact = O.run(og, case["rtype"], x, y, v, glyph=ogl, wide=True, **ch).astype(np.float64)
    assert_glyph_close(got, want, exact, case["name"], case["glyph"] == cases.LINE)


@pytest.mark.parametrize("path", PATHS)
def test_glyph_rejects_min_max(A, path):
    og = O.make_grid((0, 0, 8, 8))
    run = A.ReductionRun(cabi_grid(A, og), A.PLANE_MAX, path=path)
    try:
        with pytest.raises(A.PcrHipError) as e:
            run.scatter([1.0], [1.0], [1.0], glyph=dict(type=A.GLYPH_GAUSSIAN))
        assert e.value.code == 6 and "glyph splatting only supports" in str(e.value)
    finally:
        run.close()


@pytest.mark.parametrize("path", PATHS)
def test_glyph_quirk_probes(A, known_answers, path):
    for c in known_answers["glyph"]:
        og = grid_from_json(O, c["grid"])
        s = c["spec"]
        if s["glyph"] == "Gaussian":
            gl = dict(type=A.GLYPH_GAUSSIAN, sigma_x=s["sigma"], sigma_y=s["sigma"], max_radius=s["max_radius"])
        else:
            gl = dict(type=A.GLYPH_LINE, direction=s["direction"], half_length=s["half_length"],
                      max_radius=s["max_radius"])
        band, _ = gpu_run(A, og, RT[s["type"]], [dict(x=c["x"], y=c["y"], value=c["value"])], glyph=gl, path=path)
        for row, col, val in c.get("probes", []):
            assert band[row, col] == pytest.approx(val, rel=1e-5), (c["name"], row, col)
        if "nan_cols_from" in c:
            assert np.isnan(band[:, c["nan_cols_from"]:]).all() and np.isnan(band[c["nan_rows_from"]:, :]).all()
        if "cells_set" in c:
            assert sorted([int(r), int(cc)] for r, cc in np.argwhere(~np.isnan(band))) == sorted(c["cells_set"])


def test_row_block_shards_merge_to_single_device_result(A):
    """Multi-GPU layout on one device: two row-block shards with halo rows, halo merged with
    pcr_hip_plane_merge, must equal the unsharded result."""
    import ctypes as C
    og = O.make_grid((0, 0, 96, 64), tile=(4096, 4096))
    x, y, v = uniform_cloud(8000, og, seed=11, margin=0.0)
    gl = dict(type=A.GLYPH_GAUSSIAN, sigma_x=2.0, sigma_y=2.0, max_radius=6.0)
    ogl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=2.0, sigma_y=2.0, max_radius=6.0)
    want = O.run(og, RT["WeightedAverage"], x, y, v, glyph=ogl)
    halo, split = 6, 40
    L = A.lib()
    runs = []
    for own in ((0, split), (split, 64)):
        r = A.ReductionRun(cabi_grid(A, og, own_rows=own, halo=halo), A.PLANE_SUM | A.PLANE_WGT, path=1)
        r.scatter(x, y, v, glyph=gl)        # every shard sees the whole cloud and keeps its own rows
        runs.append(r)
    top, bot = runs
    W = og.width
    assert top.grid.state_row0 == 0 and top.grid.state_rows == split + halo
    assert bot.grid.state_row0 == split - halo
    # halo exchange: top's rows [split, split+halo) belong to bot; bot's rows [split-halo, split) to top
    for name, kind in (("d_sum", A.PLANE_SUM), ("d_wgt", A.PLANE_WGT)):
        tp, bp = top.bufs[name].ptr.value, bot.bufs[name].ptr.value
        A.check(L.pcr_hip_plane_merge(kind, C.c_void_p(bp + halo * W * 4), C.c_void_p(tp + split * W * 4), halo * W, None))
        A.check(L.pcr_hip_plane_merge(kind, C.c_void_p(tp + (split - halo) * W * 4), C.c_void_p(bp), halo * W, None))
    got = np.vstack([top.finalize(RT["WeightedAverage"]), bot.finalize(RT["WeightedAverage"])])
    st = top.stats().points_valid + bot.stats().points_valid
    for r in runs:
        r.close()
    assert st == len(x)
    assert_band_close(got, want, rtol=1e-4, atol=1e-6, what="sharded vs single")


def test_state_merge_and_init_match_oracle(A):
    L, OL = A.lib(), O.lib()
    rng = np.random.default_rng(5)
    n = 1000
    for rt in range(6):
        k = OL.pcro_state_floats(rt)
        a = rng.normal(size=k * n).astype(np.float32)
        b = rng.normal(size=k * n).astype(np.float32)
        da, db = A.DeviceBuffer.from_numpy(a), A.DeviceBuffer.from_numpy(b)
        A.check(L.pcr_hip_state_merge(rt, da.ptr, db.ptr, n, None))
        want = a.copy()
        OL.pcro_merge_state(rt, want.ctypes.data, b.ctypes.data, n)
        np.testing.assert_array_equal(da.to_numpy(), want)
        A.check(L.pcr_hip_state_init(rt, da.ptr, n, None))
        OL.pcro_init_state(rt, want.ctypes.data, n)
        np.testing.assert_array_equal(da.to_numpy(), want)


@pytest.mark.parametrize("r", [1, 2, 3, 4, 5, 6, 7, 8])
def test_default_sigma_gaussian_every_small_radius(A, r):
    """Default-sigma, unrotated Gaussians of radius r <= 7 take the separable fixed-radius splat of the LDS tiles
    (r = 8: the general wave-per-point form); anisotropic sigma, tile clipping and grid edges included."""
    og = O.make_grid((0.0, 0.0, 150.0, 110.0), tile=(64, 48))
    rng = np.random.default_rng(100 + r)
    n = 4000
    x, y = rng.uniform(-1.0, 151.0, n), rng.uniform(-1.0, 111.0, n)
    v = rng.uniform(0.5, 2.0, n).astype(np.float32)
    sx, sy = r / 3.0 - 0.01, r / 3.0 * 0.8
    gl = dict(type=A.GLYPH_GAUSSIAN, sigma_x=sx, sigma_y=sy, max_radius=float(r))
    ogl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=sx, sigma_y=sy, max_radius=float(r))
    for rname in ("WeightedAverage", "Count"):
        got, st = gpu_run(A, og, RT[rname], [dict(x=x, y=y, value=v)], glyph=gl, path=2)
        assert st.path == 1
        want = O.run(og, RT[rname], x, y, v, glyph=ogl)
        exact = O.run(og, RT[rname], x, y, v, glyph=ogl, wide=True).astype(np.float64)
        assert_glyph_close(got, want, exact, f"r={r}/{rname}", False)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("rname", ["Count", "WeightedAverage", "Sum"])
def test_gaussian_nonfinite_rotation_and_sigma_channels(A, rname, path):
    """A NaN / inf per-point rotation makes every weight of that footprint NaN (cos(NaN); NaN passes the reference's
    `w < 1e-6f` cut-off, glyph_kernels.cu:166), an inf sigma makes the footprint flat along that axis.  The LDS tiles keep the
    weight plane in 40-bit fixed point, which cannot hold a NaN: such weights must take the float path and reach the plane as
    the reference's NaN, not as a large finite number."""
    og = O.make_grid((0.0, 0.0, 160.0, 120.0), tile=(64, 48))
    rng = np.random.default_rng(77)
    n = 3000
    x, y = rng.uniform(-1.0, 161.0, n), rng.uniform(-1.0, 121.0, n)
    v = rng.uniform(0.5, 2.0, n).astype(np.float32)
    rot = rng.uniform(-3.2, 3.2, n).astype(np.float32)
    bad = rng.choice(n, 24, replace=False)
    rot[bad[:8]] = np.nan
    rot[bad[8:12]] = np.inf
    rot[bad[12:16]] = -np.inf
    sig = rng.uniform(0.5, 1.5, n).astype(np.float32)
    sig[bad[16:20]] = np.inf
    sig[bad[20:]] = np.nan                      # NaN > 0 is false: default sigma
    gl = dict(type=A.GLYPH_GAUSSIAN, sigma_x=1.2, sigma_y=0.9, max_radius=5.0)
    ogl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=1.2, sigma_y=0.9, max_radius=5.0)
    ch = dict(rotation=rot, sigma_x=sig)
    got, st = gpu_run(A, og, RT[rname], [dict(x=x, y=y, value=v, **ch)], glyph=gl, path=path)
    want = O.run(og, RT[rname], x, y, v, glyph=ogl, **ch)
    exact = O.run(og, RT[rname], x, y, v, glyph=ogl, wide=True, **ch).astype(np.float64)
    assert np.isnan(want).sum() > 24                         # the NaN footprints are there
    assert_glyph_close(got, want, exact, f"nonfinite channels/{rname}/path {path}", False)
    assert np.isfinite(got[~np.isnan(got)]).all()
