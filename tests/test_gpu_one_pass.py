"""The opt-in one-pass binning of the Point glyph (PCR_HIP_ONE_PASS=1: sampled provisioning, k_bin_sample / k_bin_provision /
k_bin_scatter1 / k_prov_items / k_ovf_apply in scatter_binned.hip; clouds of >= 2^18 points on one window of <= 4096 LDS tiles).
Exactness must not depend on the sample: every case is checked against the CPU oracle with the provisions at 100 % and
starved (PCR_HIP_DEBUG_PROVISION) so that most records take the overflow path.  Count / Max / Min bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import pcr_oracle_py as O
from conftest import assert_band_close, load_cabi

pytestmark = pytest.mark.gpu
N = 400_000


@pytest.fixture(scope="module")
def A():
    return load_cabi()


@pytest.fixture(params=[100, 35, 0], ids=["provision-100", "provision-35", "provision-0"])
def provision(request):
    old = {k: os.environ.get(k) for k in ("PCR_HIP_DEBUG_PROVISION", "PCR_HIP_ONE_PASS")}
    os.environ["PCR_HIP_DEBUG_PROVISION"] = str(request.param)
    os.environ["PCR_HIP_ONE_PASS"] = "1"                    # read when the engine is created
    yield request.param
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def clouds(kind, W, H, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        x, y = rng.uniform(-3, W + 3, n), rng.uniform(-3, H + 3, n)          # some outside
    elif kind == "sorted_y":
        x, y = rng.uniform(0, W, n), np.sort(rng.uniform(0, H, n))
    elif kind == "sorted_tile":                                               # pre-tiled input: a bin = one run of points
        x, y = rng.uniform(0, W, n), rng.uniform(0, H, n)
        order = np.lexsort((x // 128, (H - y) // 96))
        x, y = x[order], y[order]
    elif kind == "one_cell":
        x, y = np.full(n, 17.25), np.full(n, H - 40.5)
    elif kind == "hotspots":
        k = 50
        cx, cy = rng.uniform(2, W - 2, k), rng.uniform(2, H - 2, k)
        i = np.arange(n) % k
        x = np.clip(cx[i] + rng.normal(0, 1.5, n), 0, W)
        y = np.clip(cy[i] + rng.normal(0, 1.5, n), 0, H)
    else:
        raise ValueError(kind)
    v = rng.normal(0, 10, n).astype(np.float32)
    x[:3] = [np.nan, 0.0, W]                                                  # NaN coordinate, inclusive bounds
    y[:3] = [5.0, H, 0.0]
    return x, y, v


@pytest.mark.parametrize("kind", ["uniform", "sorted_y", "sorted_tile", "one_cell", "hotspots"])
def test_one_pass_matches_oracle(A, kind, provision):
    W, H = 700, 500
    x, y, v = clouds(kind, W, H, N, seed=11)
    og = O.make_grid((0, 0, W, H), tile=(256, 256))
    grid = A.make_grid((0, 0, W, H), tile=(256, 256))
    mask = A.PLANE_SUM | A.PLANE_WGT | A.PLANE_MAX | A.PLANE_MIN
    run = A.ReductionRun(grid, mask, path=A.PATH_BINNED)
    try:
        run.scatter(x, y, v)
        st = run.stats()
        assert st.path == 1 and st.scatter_chunk == 28672, "the one-pass sort was not taken"
        got = {r: run.finalize(r) for r in (A.SUM, A.COUNT, A.MAX, A.MIN, A.AVERAGE)}
    finally:
        run.close()
    ref = O.Reduction(og, O.COUNT)
    ref.ingest(x, y, v)
    assert st.points_valid == ref.points_valid()
    for r in (A.COUNT, A.MAX, A.MIN):
        assert_band_close(got[r], O.run(og, r, x, y, v), what=f"{kind}/{O.RTYPE_NAMES[r]} provision {provision}")
    for r in (A.SUM, A.AVERAGE):
        exact = O.run(og, r, x, y, v, wide=True)
        mag = np.nan_to_num(O.run(og, r, x, y, np.abs(v), wide=True))
        gn = np.isnan(got[r])
        assert np.array_equal(gn, np.isnan(exact))
        err = np.abs(got[r][~gn].astype(np.float64) - exact[~gn])
        assert (err <= 1e-5 * np.maximum(10.0, mag[~gn])).all(), f"{kind}/{O.RTYPE_NAMES[r]} provision {provision}"


def test_one_pass_second_ingest_shard_window_and_filter_mask(A, provision):
    W, H = 640, 480
    x, y, v = clouds("uniform", W, H, N, seed=5)
    og = O.make_grid((0, 0, W, H))
    keep = (np.arange(N) % 3 != 0).astype(np.uint8)                           # a point filter (FilterSpec mask)
    grid = A.make_grid((0, 0, W, H), own_rows=(100, 333))
    L = A.lib()
    run = A.ReductionRun(grid, A.PLANE_SUM | A.PLANE_WGT, path=A.PATH_BINNED)
    dmask = A.DeviceBuffer.from_numpy(keep)
    try:
        A.check(L.pcr_hip_engine_planes_fresh(run.engine, 1))                # first scatter: store-only merge, then overflow atomics
        A.check(L.pcr_hip_engine_set_point_mask(run.engine, dmask.ptr))
        run.scatter(x, y, v)
        A.check(L.pcr_hip_engine_set_point_mask(run.engine, None))
        run.scatter(x[::-1].copy(), y[::-1].copy(), v[::-1].copy())          # state accumulates
        assert run.stats().scatter_chunk == 28672
        cnt = run.finalize(A.COUNT)
    finally:
        run.close()
    k = keep.astype(bool)
    want = np.nan_to_num(O.run(og, O.COUNT, x[k], y[k], v[k])) + np.nan_to_num(O.run(og, O.COUNT, x, y, v))
    assert np.array_equal(np.nan_to_num(cnt), want[100:333])


def test_one_pass_unaligned_arrays_take_the_scalar_kernel(A, provision):
    """x / y not 16-byte aligned: every block runs the 4096-point scalar-load variant."""
    W, H = 512, 512
    x, y, v = clouds("uniform", W, H, N + 1, seed=8)
    og = O.make_grid((0, 0, W, H))
    L = A.lib()
    grid = A.make_grid((0, 0, W, H))
    run = A.ReductionRun(grid, A.PLANE_WGT | A.PLANE_MAX, path=A.PATH_BINNED)
    dx, dy, dv = A.DeviceBuffer.from_numpy(x), A.DeviceBuffer.from_numpy(y), A.DeviceBuffer.from_numpy(v)
    try:
        A.check(L.pcr_hip_scatter_point(run.engine, run.mask, C.byref(run.planes), C.c_void_p(dx.ptr.value + 8),
                                        C.c_void_p(dy.ptr.value + 8), C.c_void_p(dv.ptr.value + 4), N))
        A.check(L.pcr_hip_stream_synchronize(None))
        got_c, got_m = run.finalize(A.COUNT), run.finalize(A.MAX)
    finally:
        run.close()
    assert_band_close(got_c, O.run(og, O.COUNT, x[1:], y[1:], v[1:]), what="unaligned count")
    assert_band_close(got_m, O.run(og, O.MAX, x[1:], y[1:], v[1:]), what="unaligned max")
