"""Filter stage (first "next" row): AND of per-channel predicates.

CPU: the oracle against the reference's own known answers (tests/cpp/test_filter.cpp:31-212).
GPU: pcr_hip_filter_mask against the oracle, and the pipeline with a FilterSpec -- including the
reference's DISABLED_WithFilter expectation (tests/cpp/test_pipeline.cpp:305-360), which the
reference itself cannot pass because it routes the unfiltered cloud."""
import ctypes as C

import numpy as np
import pytest

import pcr_oracle_py as O
from conftest import assert_band_close, load_cabi

# tests/cpp/test_filter.cpp:7-29: 100 points, intensity = i, classification = i % 5
INTENSITY = np.arange(100, dtype=np.float32)
CLASSIF = (np.arange(100) % 5).astype(np.float32)
KNOWN = [   # (predicates, expected survivor count, source line)
    ([], 100, 31),
    ([("classification", "Equal", 2.0)], 20, 50),
    ([("intensity", "Less", 10.0)], 10, 71),
    ([("intensity", "GreaterEqual", 90.0)], 10, 91),
    ([("classification", "InSet", [1.0, 3.0])], 40, 111),
    ([("classification", "NotInSet", [0.0, 1.0])], 60, 132),
    ([("intensity", "GreaterEqual", 50.0), ("intensity", "Less", 60.0), ("classification", "Equal", 0.0)], 2, 157),
    ([("intensity", "Greater", 1000.0)], 0, 186),
]
CH = {"intensity": INTENSITY, "classification": CLASSIF}


@pytest.mark.parametrize("preds,count,line", KNOWN, ids=[f"test_filter.cpp:{k[2]}" for k in KNOWN])
def test_oracle_filter_known_answers(preds, count, line):
    mask, kept = O.filter_mask(100, [(CH[c], op, val) for c, op, val in preds])
    assert kept == count and int(mask.sum()) == count
    if preds == KNOWN[6][0]:
        assert np.flatnonzero(mask).tolist() == [50, 55]


def test_oracle_filter_nan_semantics():
    v = np.array([np.nan, 1.0, 2.0], dtype=np.float32)
    assert O.filter_mask(3, [(v, "NotEqual", 1.0)])[0].tolist() == [1, 0, 1]     # NaN != x is true
    assert O.filter_mask(3, [(v, "Less", 5.0)])[0].tolist() == [0, 1, 1]         # NaN < x is false
    assert O.filter_mask(3, [(v, "NotInSet", [1.0])])[0].tolist() == [1, 0, 1]


def _gpu_mask(A, n, preds):
    L = A.lib()
    keep = []
    arr = (A.Predicate * max(len(preds), 1))()
    for k, (ch, op, val) in enumerate(preds):
        b = A.DeviceBuffer.from_numpy(np.ascontiguousarray(ch, dtype=np.float32))
        keep.append(b)
        arr[k].d_channel = b.ptr.value
        arr[k].op = O.CMP[op]
        if op in ("InSet", "NotInSet"):
            arr[k].set_size = len(val)
            for j, s in enumerate(val):
                arr[k].set[j] = s
        else:
            arr[k].value = val
    dmask = A.DeviceBuffer(max(n, 1))
    dcount = A.DeviceBuffer(8)
    A.check(L.pcr_hip_filter_mask(arr, len(preds), n, dmask.ptr, dcount.ptr, None))
    A.check(L.pcr_hip_stream_synchronize(None))
    return dmask.to_numpy(np.uint8, (n,)), int(dcount.to_numpy(np.uint64, (1,))[0]), dmask, keep


@pytest.mark.gpu
def test_gpu_filter_mask_matches_oracle():
    A = load_cabi()
    for preds, count, _ in KNOWN:
        mask, kept, _, _ = _gpu_mask(A, 100, [(CH[c], op, val) for c, op, val in preds])
        want, wk = O.filter_mask(100, [(CH[c], op, val) for c, op, val in preds])
        assert kept == count == wk and np.array_equal(mask, want)
    rng = np.random.default_rng(0)
    n = 200_003
    a = rng.normal(size=n).astype(np.float32)
    a[::97] = np.nan
    b = rng.integers(0, 9, n).astype(np.float32)
    preds = [(a, "Greater", -0.5), (a, "LessEqual", 1.25), (b, "NotInSet", [2.0, 5.0, 7.0]), (b, "NotEqual", 0.0)]
    mask, kept, _, _ = _gpu_mask(A, n, preds)
    want, wk = O.filter_mask(n, preds)
    assert kept == wk and np.array_equal(mask, want)


@pytest.mark.gpu
@pytest.mark.parametrize("path", [1, 2])
def test_gpu_scatter_honours_point_mask(path):
    A = load_cabi()
    og = O.make_grid((0, 0, 64, 48))
    rng = np.random.default_rng(4)
    n = 20000
    x, y = rng.uniform(0, 64, n), rng.uniform(0, 48, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    cls = rng.integers(0, 4, n).astype(np.float32)
    mask, kept, dmask, keep = _gpu_mask(A, n, [(cls, "InSet", [1.0, 2.0])])
    run = A.ReductionRun(A.make_grid((0, 0, 64, 48)), A.PLANE_SUM | A.PLANE_WGT, path=path)
    try:
        A.check(A.lib().pcr_hip_engine_set_point_mask(run.engine, dmask.ptr))
        run.scatter(x, y, v)
        assert run.stats().points_valid == kept
        got = run.finalize(A.AVERAGE)
        gl = dict(type=A.GLYPH_GAUSSIAN, sigma_x=1.5, sigma_y=1.5, max_radius=5.0)
        run2 = A.ReductionRun(A.make_grid((0, 0, 64, 48)), 3, path=path, engine=run.engine)
        run2.scatter(x, y, v, glyph=gl)
        got_g = run2.finalize(A.WEIGHTED_AVERAGE)
        run2.close()
    finally:
        run.close()
    m = mask.astype(bool)
    assert_band_close(got, O.run(og, O.AVERAGE, x[m], y[m], v[m]), rtol=1e-5, atol=1e-6, what="masked point")
    want_g = O.run(og, O.WEIGHTED_AVERAGE, x[m], y[m], v[m],
                   glyph=O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=1.5, sigma_y=1.5, max_radius=5.0))
    assert_band_close(got_g, want_g, rtol=1e-4, atol=1e-6, what="masked gaussian")


@pytest.mark.gpu
def test_pipeline_with_filter_reference_expectation():
    """tests/cpp/test_pipeline.cpp:305-360 (DISABLED_WithFilter): 100 points, one per cell of the 10x10
    grid, classification = idx % 2, filter classification == 1, Count -> total 50."""
    import pcr
    cfg = pcr.PipelineConfig()
    cfg.grid.bounds = pcr.BBox(0.0, 0.0, 10.0, 10.0)
    cfg.grid.tile_width = cfg.grid.tile_height = 5
    cfg.grid.compute_dimensions()
    cfg.exec_mode = pcr.ExecutionMode.GPU
    r = pcr.ReductionSpec()
    r.value_channel, r.type = "intensity", pcr.ReductionType.Count
    cfg.reductions = [r]
    f = pcr.FilterSpec()
    f.add("classification", pcr.CompareOp.Equal, 1.0)
    cfg.filter = f
    pipe = pcr.Pipeline.create(cfg)
    assert pipe is not None
    idx = np.arange(100)
    c = pcr.PointCloud.create(100)
    c.set_x_array(0.5 + (idx % 10))
    c.set_y_array(9.5 - (idx // 10))
    c.add_channel("intensity", pcr.DataType.Float32)
    c.set_channel_array_f32("intensity", np.ones(100, dtype=np.float32))
    c.add_channel("classification", pcr.DataType.Float32)
    c.set_channel_array_f32("classification", (idx % 2).astype(np.float32))
    for cloud in (c, c.to_device()):
        pipe.ingest(cloud)
    pipe.finalize()
    band = np.array(pipe.result().band_array(0))
    assert np.nansum(band) == 100                       # two ingests x 50 survivors
    assert pipe.stats().points_processed == 100         # points_processed counts survivors (pipeline.cpp:749)
    expect = np.where((idx % 2).reshape(10, 10) == 1, 2.0, np.nan).astype(np.float32)
    assert_band_close(band, expect, what="WithFilter")
    # a filter nothing passes is a successful no-op (pipeline.cpp:349-353)
    f2 = pcr.FilterSpec()
    f2.add("classification", pcr.CompareOp.Greater, 5.0)
    cfg.filter = f2
    p2 = pcr.Pipeline.create(cfg)
    p2.ingest(c)
    assert p2.stats().points_processed == 0 and p2.stats().collections_processed == 0
