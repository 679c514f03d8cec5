"""Kernel variants that the default benchmark shapes do not reach, against the CPU oracle:
  * the matrix-core column pass of the moment path (k_conv_col_mfma) on widths that are not a multiple of 4 (scalar staging),
    odd radii (slack rows), radii that need the 14-round and the unrolled-less staging variants;
  * the moment path at K = 3, 4, 5 on shapes with tiles and odd widths (round 5 removed the last experiment switch,
    PCR_HIP_TUNE_CONV: each column-pass kernel is tested at the radii the engine runs it at, r < 24 vector ALU, r >= 24 matrix cores).
Same tolerance as every Gaussian / Line test (rtol 1e-4, exact NaN mask)."""
import os

import numpy as np
import pytest

import pcr_oracle_py as O
from conftest import load_cabi
from test_gpu_moments import RT, check, run_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    return load_cabi()


class env:
    """Engine knobs are read when the engine is created: set for the duration of one run."""

    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def gaussian_case(A, G, sigma, maxr, n, tile=(4096, 4096), rname="WeightedAverage", seed=23):
    W, H = G
    og = O.make_grid((0.0, 0.0, float(W), float(H)), tile=tile)
    rng = np.random.default_rng(seed)
    x = rng.uniform(-2.0, W + 2.0, n)
    y = rng.uniform(-2.0, H + 2.0, n)
    v = rng.normal(10.0, 3.0, n).astype(np.float32)
    gl = dict(type=A.GLYPH_GAUSSIAN, sigma_x=sigma, sigma_y=sigma, max_radius=maxr)
    ogl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=sigma, sigma_y=sigma, max_radius=maxr)
    rt = RT[rname]
    got, st, run = run_gpu(A, og, rt, x, y, v, gl, path=3)
    run.close()
    assert st.path == 2, "moment path was not taken"
    want = O.run(og, rt, x, y, v, glyph=ogl)
    exact = O.run(og, rt, x, y, v, glyph=ogl, wide=True).astype(np.float64)
    check(got, want, exact, f"G={G} sigma={sigma} r<={maxr}", scale=1.0 if rname == "Count" else 10.0)


MFMA_SHAPES = [
    dict(id="w203_scalar_staging", G=(203, 150), sigma=8.0, maxr=24.0, n=5000),            # W % 4 != 0
    dict(id="w1001_scalar_staging_tiles", G=(1001, 96), sigma=8.0, maxr=30.0, n=6000, tile=(300, 64)),
    dict(id="r25_odd_radius", G=(256, 200), sigma=9.0, maxr=25.0, n=5000),                 # 2r = 2 (mod 4): slack rows
    dict(id="r27_odd_radius_tiles", G=(300, 260), sigma=9.0, maxr=40.0, n=5000, tile=(128, 100)),
    dict(id="r60_14_rounds", G=(320, 300), sigma=20.0, maxr=60.0, n=4000),
    dict(id="r100_loop_staging", G=(420, 400), sigma=34.0, maxr=100.0, n=3000),
    dict(id="r36_short_window", G=(256, 40), sigma=12.0, maxr=64.0, n=3000),                # window shorter than the tap support
    dict(id="r190_largest_lds_image", G=(430, 410), sigma=64.0, maxr=190.0, n=700),         # 464 staged rows: 155 KB of LDS
]


@pytest.mark.parametrize("case", MFMA_SHAPES, ids=lambda c: c["id"])
@pytest.mark.parametrize("rname", ["WeightedAverage", "Count"])
def test_matrix_core_column_pass_shapes(A, case, rname):
    gaussian_case(A, case["G"], case["sigma"], case["maxr"], case["n"], tile=case.get("tile", (4096, 4096)), rname=rname)


@pytest.mark.parametrize("shape", [dict(G=(300, 230), sigma=16.0, maxr=48.0, n=6000),               # K = 3, matrix cores
                                   dict(G=(260, 200), sigma=4.0, maxr=12.0, n=8000, tile=(128, 96)),  # K = 5, r = 12, vector ALU
                                   dict(G=(1301, 90), sigma=8.0, maxr=24.0, n=6000)],                 # K = 4, r = 24: the first matrix-core radius
                         ids=["s16", "s4_tiles", "s8_w1301"])
def test_moment_path_orders_and_column_kernels(A, shape):
    gaussian_case(A, shape["G"], shape["sigma"], shape["maxr"], shape["n"], tile=shape.get("tile", (4096, 4096)))


def test_line_records_count_and_sum_exact(A):
    """Line tiles on 16-byte end-point records (bin16.hpp): Count bit-exact, Sum within tolerance, on a grid with several
    LDS tiles per side and segments that cross them."""
    W, H, n = 700, 500, 60000
    og = O.make_grid((0.0, 0.0, float(W), float(H)))
    rng = np.random.default_rng(4)
    x, y = rng.uniform(-3, W + 3, n), rng.uniform(-3, H + 3, n)
    v = rng.uniform(1.0, 10.0, n).astype(np.float32)         # one sign: the check below is relative
    d = rng.uniform(0, np.pi, n).astype(np.float32)
    gl = dict(type=A.GLYPH_LINE, half_length=9.0, max_radius=11.0)
    ogl = O.make_glyph(O.GLYPH_LINE, half_length=9.0, max_radius=11.0)
    grid = A.make_grid((0.0, 0.0, float(W), float(H)), dims=(W, H))
    results = {}
    for rname, mask in (("Count", A.PLANE_WGT), ("Sum", A.PLANE_SUM)):
        run = A.ReductionRun(grid, mask, path=2)
        try:
            run.scatter(x, y, v, glyph=gl, direction=d)
            results[rname] = run.finalize(RT[rname])
            assert run.stats().path == 1
        finally:
            run.close()
    want = O.run(og, RT["Count"], x, y, v, glyph=ogl, direction=d)
    assert np.array_equal(np.nan_to_num(results["Count"], nan=-1.0), np.nan_to_num(want, nan=-1.0))
    want_s = O.run(og, RT["Sum"], x, y, v, glyph=ogl, direction=d)
    exact = O.run(og, RT["Sum"], x, y, v, glyph=ogl, direction=d, wide=True).astype(np.float64)
    check(results["Sum"], want_s, exact, "line sum", scale=5.0)


def test_line_segments_beyond_the_record_range_take_the_list(A):
    """A per-point half_length of tens of thousands of cells does not fit a record's int16 end-point offsets: such
    segments go to the list and are walked by the direct form (clipped to their reference tile as ever).  Count
    bit-exact against the oracle."""
    W, H, n = 300, 200, 5000
    og = O.make_grid((0.0, 0.0, float(W), float(H)), tile=(128, 64))
    rng = np.random.default_rng(8)
    x, y = rng.uniform(0, W, n), rng.uniform(0, H, n)
    v = rng.uniform(1.0, 2.0, n).astype(np.float32)
    d = rng.uniform(0, np.pi, n).astype(np.float32)
    hl = rng.uniform(0.5, 6.0, n).astype(np.float32)
    hl[::97] = 60000.0                                       # far beyond +-32000 cells
    gl = dict(type=A.GLYPH_LINE, half_length=2.0, max_radius=1e9)
    ogl = O.make_glyph(O.GLYPH_LINE, half_length=2.0, max_radius=1e9)
    grid = A.make_grid((0.0, 0.0, float(W), float(H)), dims=(W, H), tile=(128, 64))
    run = A.ReductionRun(grid, A.PLANE_WGT, path=2)
    try:
        run.scatter(x, y, v, glyph=gl, direction=d, half_length=hl)
        got = run.finalize(RT["Count"])
        assert run.stats().path == 1
    finally:
        run.close()
    want = O.run(og, RT["Count"], x, y, v, glyph=ogl, direction=d, half_length=hl)
    assert np.array_equal(np.nan_to_num(got, nan=-1.0), np.nan_to_num(want, nan=-1.0))


def _large_gaussian_case(seed):
    rng = np.random.default_rng(9000 + seed)
    W, H = int(rng.integers(60, 700)), int(rng.integers(60, 520))
    cs = float(rng.choice([0.5, 1.0, 2.0]))
    tile = (int(rng.choice([96, 200, 257, 4096])), int(rng.choice([80, 128, 301, 4096])))
    og = O.make_grid((0.0, 0.0, W * cs, H * cs), cell=(cs, -cs), tile=tile)
    n = int(rng.integers(300, 5000))
    x = rng.uniform(-3 * cs, og.max_x + 3 * cs, n)
    y = rng.uniform(-3 * cs, og.max_y + 3 * cs, n)
    if rng.uniform() < 0.5:                                      # half of the cloud in one blob
        k = n // 2
        x[:k] = rng.normal(0.4 * og.max_x, 3 * cs, k)
        y[:k] = rng.normal(0.5 * og.max_y, 3 * cs, k)
    v = rng.uniform(1.0, 20.0, n).astype(np.float32)
    s = float(rng.uniform(8.0, 17.0))
    sx, sy = s * cs, s * cs * float(rng.choice([1.0, 1.0, 0.9, 1.1]))
    maxr = float(rng.choice([24.0, 25.0, 31.0, 40.0, 48.0, 64.0]))
    rname = str(rng.choice(["WeightedAverage", "Sum", "Count"]))
    return og, x, y, v, sx, sy, maxr, rname


@pytest.mark.parametrize("seed", range(36))
def test_random_large_gaussians_on_the_matrix_core_pass(A, seed):
    """Random grids, reference tiles, radii 24..51 and widths of any parity through the moment path (r >= 24: k_conv_col_mfma)."""
    og, x, y, v, sx, sy, maxr, rname = _large_gaussian_case(seed)
    gl = dict(type=A.GLYPH_GAUSSIAN, sigma_x=sx, sigma_y=sy, max_radius=maxr)
    ogl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=sx, sigma_y=sy, max_radius=maxr)
    rt = RT[rname]
    try:
        got, st, run = run_gpu(A, og, rt, x, y, v, gl, path=3)
    except A.PcrHipError:
        pytest.skip("expansion not applicable to this spec (cut-off condition)")
    run.close()
    assert st.path == 2
    want = O.run(og, rt, x, y, v, glyph=ogl)
    exact = O.run(og, rt, x, y, v, glyph=ogl, wide=True).astype(np.float64)
    check(got, want, exact, f"seed {seed}: {og.width}x{og.height} tile {og.tile_width}x{og.tile_height} "
          f"sigma {sx:.2f},{sy:.2f} r<={maxr} {rname}", scale=1.0 if rname == "Count" else 10.0)


def test_line_tiles_count_field_holds_a_full_item(A):
    """The packed Line window counts a cell's visits in 16 bits: 70 000 identical segments (two work items, the first with the
    65 535 records an item may hold) put 65 535 visits on every cell of the segment without carrying into nothing."""
    W, H, n = 200, 200, 70000
    og = O.make_grid((0.0, 0.0, float(W), float(H)))
    rng = np.random.default_rng(5)
    x, y = np.full(n, 100.3), np.full(n, 90.6)
    d = np.full(n, 0.4, dtype=np.float32)
    v = rng.uniform(0.5, 1.0, n).astype(np.float32)
    gl = dict(type=A.GLYPH_LINE, half_length=6.0, max_radius=8.0)
    ogl = O.make_glyph(O.GLYPH_LINE, half_length=6.0, max_radius=8.0)
    grid = A.make_grid((0.0, 0.0, float(W), float(H)), dims=(W, H))
    run = A.ReductionRun(grid, 3, path=2)
    try:
        run.scatter(x, y, v, glyph=gl, direction=d)
        got_s, got_c = run.plane("d_sum").astype(np.float64), run.plane("d_wgt")
        assert run.stats().path == 1
    finally:
        run.close()
    want_c = np.nan_to_num(O.run(og, RT["Count"], x, y, v, glyph=ogl, direction=d))
    assert want_c.max() == n and np.array_equal(got_c, want_c)
    exact = O.run(og, RT["Sum"], x, y, v, glyph=ogl, direction=d, wide=True).astype(np.float64)
    assert np.allclose(got_s, exact, rtol=4e-7, atol=0.0)


@pytest.mark.parametrize("spread", ["one_exponent", "seven_exponents", "wide", "wide_dense", "nonfinite"])
def test_line_tiles_packed_and_f64_forms_agree_with_the_oracle(A, spread):
    """Line tiles sum values in f64 inside their LDS window and round once at the merge: a cell whose values are all tiny next to
    its neighbours' keeps its full relative precision (the check is relative to the cell's own sum of |v|), whatever the spread
    of exponents inside a tile, and NaN / inf values stay where the reference puts them.  (Round 3 tried two integer forms of
    the window -- visits and value packed in one 64-bit atomic; the value as exact 64-bit fixed point with an f64 redo -- both
    green on this test, neither faster: the tile kernel was bound by its walk then.  Round 4's kernel IS the packed form for
    two planes: one ds_add_u64 per visited cell, an eight-exponent window, the values outside it listed in LDS and walked as
    doubles afterwards -- "wide" lists ~half of an item's segments, "wide_dense" more than the list holds: the item is scanned
    again for them.  DESIGN section 3a.)"""
    W, H, n = 400, 300, (400000 if spread == "wide_dense" else 40000)
    og = O.make_grid((0.0, 0.0, float(W), float(H)))
    rng = np.random.default_rng(31)
    x, y = rng.uniform(0, W, n), rng.uniform(0, H, n)
    d = rng.uniform(0, np.pi, n).astype(np.float32)
    if spread == "one_exponent":
        v = rng.uniform(1.0, 2.0, n)
    elif spread == "seven_exponents":
        v = rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-3, 5, n) * rng.choice([-1.0, 1.0], n)    # exponents -3 .. 4, both signs
    else:
        v = rng.uniform(1.0, 2.0, n) * np.where(x < W / 2, 1e-7, 1e4)     # 37 binary exponents apart inside most tiles
    v = v.astype(np.float32)
    v[::1000] = 0.0                                                        # zeros do not count towards the range
    if spread == "nonfinite":
        v[5], v[6] = np.inf, np.nan
    gl = dict(type=A.GLYPH_LINE, half_length=6.0, max_radius=8.0)
    ogl = O.make_glyph(O.GLYPH_LINE, half_length=6.0, max_radius=8.0)
    grid = A.make_grid((0.0, 0.0, float(W), float(H)), dims=(W, H))
    run = A.ReductionRun(grid, 3, path=2)
    try:
        run.scatter(x, y, v, glyph=gl, direction=d)
        got_s, got_c = run.plane("d_sum").astype(np.float64), run.plane("d_wgt")
        assert run.stats().path == 1
    finally:
        run.close()
    want_c = np.nan_to_num(O.run(og, RT["Count"], x, y, v, glyph=ogl, direction=d))
    assert np.array_equal(got_c, want_c)
    exact = O.run(og, RT["Sum"], x, y, v, glyph=ogl, direction=d, wide=True).astype(np.float64)
    mag = np.nan_to_num(O.run(og, RT["Sum"], x, y, np.abs(np.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0)), glyph=ogl,
                              direction=d, wide=True).astype(np.float64))
    assert np.array_equal(np.isfinite(exact), np.isfinite(got_s)) and np.array_equal(np.isnan(exact), np.isnan(got_s))
    occ = np.isfinite(exact) & (want_c > 0)
    err = np.abs(got_s[occ] - exact[occ])
    # f32 merges of a few window partial sums per cell: a few ulp of the cell's own magnitude
    assert (err <= 4e-7 * np.maximum(mag[occ], 1e-30)).all(), f"{spread}: max rel err {np.max(err / np.maximum(mag[occ], 1e-30)):.3e}"


@pytest.mark.parametrize("dims,shape", [((2048, 1024), "512x24 (<= 2048 bins)"), ((8192, 4096), "1024x16, one staging round (2752 bins)"),
                                        ((8192, 8192), "1024x16, 8192-record window (5504 bins)")],
                         ids=["few_bins", "mid_bins", "many_bins"])
def test_point_scatter_shapes_by_bin_count(A, dims, shape):
    """The Point scatter pass picks its workgroup shape from the number of LDS tiles (scatter_shape, scatter_binned.hip): one
    case per shape, with a ragged end of the cloud (n is no multiple of any chunk) -- Count bit-exact, Sum and Max against the
    oracle."""
    W, H = dims
    n = 3_000_017
    rng = np.random.default_rng(W + H)
    x, y = rng.uniform(0, W, n), rng.uniform(0, H, n)
    v = rng.uniform(-1, 1, n).astype(np.float32)
    og = O.make_grid((0.0, 0.0, float(W), float(H)))
    grid = A.make_grid((0.0, 0.0, float(W), float(H)), dims=(W, H))
    run = A.ReductionRun(grid, 7, path=2)                     # sum + count + max planes
    try:
        run.scatter(x, y, v)
        st = run.stats()
        assert st.path == 1 and st.num_bins == -(-W // st.lds_tile_w) * -(-H // st.lds_tile_h), shape
        got_s, got_c, got_m = run.plane("d_sum"), run.plane("d_wgt"), run.plane("d_max")
    finally:
        run.close()
    want_c = np.nan_to_num(O.run(og, O.COUNT, x, y, v))
    assert np.array_equal(got_c, want_c), shape
    want_s = np.nan_to_num(O.run(og, O.SUM, x, y, v, wide=True).astype(np.float64))
    assert (np.abs(got_s - want_s) <= 1e-5 * np.maximum(1.0, np.abs(want_s))).all(), shape
    want_m = O.run(og, O.MAX, x, y, v)
    occ = want_c > 0
    assert np.array_equal(got_m[occ], want_m[occ]), shape


@pytest.mark.parametrize("n,two_level", [(80_000_123, False), (6145 * 12288 + 5000, False), (6145 * 12288 + 5000, True)],
                         ids=["6510_blocks", "6145_blocks_last_group_past_the_end", "6145_blocks_two_level"])
def test_count_pass_with_several_scatter_blocks_per_workgroup(A, n, two_level, monkeypatch):
    """Few bins and many points: the count pass gives a workgroup several scatter blocks of one virtual XCD (k_bin_count<true>:
    80 M points on a 1024^2 grid, 88 LDS tiles, 6 511 scatter blocks -> two per workgroup), ragged end included -- Count
    bit-exact and conserved, Sum against the oracle.  6 145 full blocks (6145 % 8 = 1, (769 - 1) % 2 = 0): the last group of
    count workgroups starts at blocks 6144 .. 6151, seven of which do not exist -- round 4's kernel let those seven count the
    ragged end a second time; also as the first level of the two-level sort."""
    if two_level:
        monkeypatch.setenv("PCR_HIP_DEBUG_MAX_BINS", "24")          # read by pcr_hip_engine_create: 88 tiles -> two sort levels
        monkeypatch.setenv("PCR_HIP_DEBUG_TWO_LEVEL", "1")
    G = 1024
    rng = np.random.default_rng(99)
    x, y = rng.uniform(0, G, n), rng.uniform(0, G, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    og = O.make_grid((0.0, 0.0, float(G), float(G)))
    grid = A.make_grid((0.0, 0.0, float(G), float(G)), dims=(G, G))
    run = A.ReductionRun(grid, 3, path=2)
    try:
        run.scatter(x, y, v)
        assert run.stats().path == 1 and run.stats().points_valid == n
        got_s, got_c = run.plane("d_sum"), run.plane("d_wgt")
    finally:
        run.close()
    want_c = np.nan_to_num(O.run(og, O.COUNT, x, y, v))
    assert np.array_equal(got_c, want_c) and got_c.astype(np.float64).sum() == n
    want_s = np.nan_to_num(O.run(og, O.SUM, x, y, v, wide=True).astype(np.float64))
    assert (np.abs(got_s - want_s) <= 1e-5 * np.maximum(1.0, np.abs(want_s))).all()
