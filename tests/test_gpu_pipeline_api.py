"""The drop-in Python API (pcr.Pipeline & co.) on the GPU, against the reference's own pipeline
known answers and the CPU oracle.  These read like tests/cpp/test_pipeline.cpp of the reference."""
import numpy as np
import pytest

import pcr
import pcr_oracle_py as O
from conftest import assert_band_close, grid_from_json

pytestmark = pytest.mark.gpu

RT = {"Sum": pcr.ReductionType.Sum, "Max": pcr.ReductionType.Max, "Min": pcr.ReductionType.Min,
      "Average": pcr.ReductionType.Average, "WeightedAverage": pcr.ReductionType.WeightedAverage,
      "Count": pcr.ReductionType.Count}
ORT = {"Sum": O.SUM, "Max": O.MAX, "Min": O.MIN, "Average": O.AVERAGE,
       "WeightedAverage": O.WEIGHTED_AVERAGE, "Count": O.COUNT}


def config_for(og, reductions, **kw):
    cfg = pcr.PipelineConfig()
    cfg.grid.bounds = pcr.BBox(og.min_x, og.min_y, og.max_x, og.max_y)
    cfg.grid.cell_size_x, cfg.grid.cell_size_y = og.cell_size_x, og.cell_size_y
    cfg.grid.tile_width, cfg.grid.tile_height = og.tile_width, og.tile_height
    cfg.grid.width, cfg.grid.height = og.width, og.height
    cfg.grid.tiles_x = -(-og.width // og.tile_width)
    cfg.grid.tiles_y = -(-og.height // og.tile_height)
    cfg.exec_mode = pcr.ExecutionMode.GPU
    cfg.reductions = reductions
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def spec(rtype, channel="value"):
    r = pcr.ReductionSpec()
    r.value_channel, r.type = channel, RT[rtype]
    return r


def cloud_from(x, y, channels, loc="host"):
    c = pcr.PointCloud.create(max(len(x), 1))
    c.set_x_array(np.asarray(x, dtype=np.float64))
    c.set_y_array(np.asarray(y, dtype=np.float64))
    c.resize(len(x))
    for name, arr in channels.items():
        c.add_channel(name, pcr.DataType.Float32)
        if len(x):
            c.set_channel_array_f32(name, np.asarray(arr, dtype=np.float32))
    return c.to_device() if loc == "device" else c


@pytest.mark.parametrize("loc", ["host", "device"])
def test_reference_pipeline_known_answers(known_answers, denan, loc):
    for case in known_answers["pipeline"]:
        og = grid_from_json(O, case["grid"])
        cfg = config_for(og, [spec(r["type"]) for r in case["reductions"]])
        pipe = pcr.Pipeline.create(cfg)
        assert pipe is not None, pcr.pipeline_create_error()
        pipe.validate()
        for cl in case["clouds"]:
            pipe.ingest(cloud_from(cl["x"], cl["y"], {"value": cl["value"]}, loc))
        pipe.finalize()
        res = pipe.result()
        assert (res.cols(), res.rows(), res.num_bands()) == (og.width, og.height, len(case["reductions"]))
        for b in range(res.num_bands()):
            want = np.array(denan(case["expected"][b]), dtype=np.float32).reshape(og.height, og.width)
            assert_band_close(res.band_array(b), want, what=f'{case["name"]} band {b}')


def test_band_names_stats_and_run():
    # band naming pipeline.cpp:1175-1186; run() + stats() test_pipeline.cpp:398-441
    og = O.make_grid((0, 0, 10, 10), tile=(5, 5))
    r2 = spec("Count")
    r2.output_band_name = "n"
    pipe = pcr.Pipeline.create(config_for(og, [spec("Sum"), r2]))
    c1 = cloud_from([0.5, 1.5], [9.5, 9.5], {"value": [1, 2]})
    c2 = cloud_from([7.5], [2.5], {"value": [5]})
    pipe.run([c1, c2])
    res = pipe.result()
    assert res.band_desc(0).name == "value_0" and res.band_desc(1).name == "n"
    st = pipe.stats()
    assert st.collections_processed == 2 and st.points_processed == 3 and st.tiles_active == 2
    assert st.collections_total == 0 and st.elapsed_seconds >= 0
    assert res.band_array(0)[0, 0] == 1 and res.band_array(1)[7, 7] == 1
    assert np.isnan(res.band_array(0)[9, 0])          # untouched tile


def test_progress_callback_and_cancel():
    # test_pipeline.cpp:443-480 ; pipeline.cpp:753-767
    og = O.make_grid((0, 0, 10, 10))
    pipe = pcr.Pipeline.create(config_for(og, [spec("Sum")]))
    seen = []
    pipe.set_progress_callback(lambda info: seen.append((info.collections_processed, info.points_processed)) or True)
    c = cloud_from([1.0, 2.0], [1.0, 2.0], {"value": [1, 1]})
    pipe.ingest(c)
    pipe.ingest(c)
    assert seen == [(1, 2), (2, 4)]
    pipe.set_progress_callback(lambda info: False)
    with pytest.raises(RuntimeError, match="cancelled by user"):
        pipe.ingest(c)


def test_error_messages_match_reference():
    og = O.make_grid((0, 0, 10, 10))
    pipe = pcr.Pipeline.create(config_for(og, [spec("Sum", "intensity")]))
    c = cloud_from([1.0], [1.0], {"value": [1]})
    with pytest.raises(RuntimeError, match="pipeline: value channel not found: intensity"):
        pipe.ingest(c)
    c.add_channel("intensity", pcr.DataType.Int32)
    with pytest.raises(RuntimeError, match="pipeline: value channel must be Float32"):
        pipe.ingest(c)
    # glyph + Max -> NotImplemented text of pipeline.cpp:500-508
    s = pcr.gaussian_splat_spec("value")
    s.type = pcr.ReductionType.Max
    pipe = pcr.Pipeline.create(config_for(og, [s]))
    with pytest.raises(RuntimeError, match="glyph splatting only supports"):
        pipe.ingest(cloud_from([1.0], [1.0], {"value": [1]}))
    # empty pipeline: validate fails, create succeeds (test_pipeline.cpp:57-64)
    pipe = pcr.Pipeline.create(config_for(og, []))
    assert pipe is not None
    with pytest.raises(RuntimeError, match="at least one reduction"):
        pipe.validate()
    # filter on a missing / non-Float32 channel: filter_points' messages (src/engine/filter.cpp:101-123)
    cfg = config_for(og, [spec("Sum")])
    f = pcr.FilterSpec()
    f.add("classification", pcr.CompareOp.Greater, 0.5)
    cfg.filter = f
    pipe = pcr.Pipeline.create(cfg)
    c = cloud_from([1.0], [1.0], {"value": [1]})
    with pytest.raises(RuntimeError, match="filter_points: channel not found: classification"):
        pipe.ingest(c)
    c.add_channel("classification", pcr.DataType.Int32)
    with pytest.raises(RuntimeError, match="only Float32 channels supported for filtering"):
        pipe.ingest(c)
    # output_path that cannot be created: the result is still produced, the write error surfaces
    cfg = config_for(og, [spec("Sum")], output_path="/nonexistent-dir/out.tif")
    pipe = pcr.Pipeline.create(cfg)
    pipe.ingest(cloud_from([1.0], [1.0], {"value": [1]}))
    with pytest.raises(RuntimeError, match="failed to create GeoTIFF"):
        pipe.finalize()
    assert pipe.result().band_array(0)[9, 1] == 1.0


@pytest.mark.parametrize("path", [1, 0])
def test_c1_shape_point_average_1000sq(path):
    """BASELINE config[0] at reduced N: uniform points, 1000x1000 grid, Point glyph, Average."""
    G, n = 1000, 300_000
    rng = np.random.default_rng(42)
    x, y = rng.uniform(2, G - 2, n), rng.uniform(2, G - 2, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    og = O.make_grid((0, 0, G, G))
    pipe = pcr.Pipeline.create(config_for(og, [spec("Average")], scatter_path=path))
    pipe.ingest(cloud_from(x, y, {"value": v}, "device"))
    pipe.finalize()
    got = np.array(pipe.result().band_array(0))
    assert_band_close(got, O.run(og, O.AVERAGE, x, y, v), rtol=1e-5, atol=1e-6, what="C1")
    assert pipe.last_scatter()["points_valid"] == n


@pytest.mark.parametrize("path", [1, 0])
def test_mixed_reductions_share_passes(path):
    """Sum+Count+Average (C2's reduction set) + Min+Max on a second channel + two glyph specs."""
    G, n = 384, 120_000
    rng = np.random.default_rng(5)
    x, y = rng.uniform(0, G, n), rng.uniform(0, G, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    z = rng.normal(100, 30, n).astype(np.float32)
    d = rng.uniform(0, np.pi, n).astype(np.float32)
    og = O.make_grid((0, 0, G, G), tile=(128, 128))
    specs = [spec("Sum"), spec("Count"), spec("Average"), spec("Min", "z"), spec("Max", "z"),
             pcr.gaussian_splat_spec("z", default_sigma=1.5, max_radius_cells=5.0),
             pcr.line_splat_spec("value", direction_channel="dir", default_half_length=6.0, max_radius_cells=8.0)]
    pipe = pcr.Pipeline.create(config_for(og, specs, scatter_path=path))
    pipe.ingest(cloud_from(x, y, {"value": v, "z": z, "dir": d}, "device"))
    pipe.finalize()
    res = pipe.result()
    want = [O.run(og, O.SUM, x, y, v), O.run(og, O.COUNT, x, y, v), O.run(og, O.AVERAGE, x, y, v),
            O.run(og, O.MIN, x, y, z), O.run(og, O.MAX, x, y, z),
            O.run(og, O.WEIGHTED_AVERAGE, x, y, z, glyph=O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=1.5, sigma_y=1.5, max_radius=5.0)),
            O.run(og, O.WEIGHTED_AVERAGE, x, y, v, glyph=O.make_glyph(O.GLYPH_LINE, half_length=6.0, max_radius=8.0), direction=d)]
    tols = [(1e-5, 1e-5), (0, 0), (1e-5, 1e-6), (0, 0), (0, 0), (1e-4, 1e-5), (1e-4, 1e-6)]
    for b, (w, (rt, at)) in enumerate(zip(want, tols)):
        assert_band_close(np.array(res.band_array(b)), w, rtol=rt, atol=at, what=f"band {b}")


def test_device_resident_result_and_host_copy():
    og = O.make_grid((0, 0, 64, 32))
    rng = np.random.default_rng(1)
    x, y = rng.uniform(0, 64, 5000), rng.uniform(0, 32, 5000)
    v = rng.uniform(0, 1, 5000).astype(np.float32)
    pipe = pcr.Pipeline.create(config_for(og, [spec("Count")], result_location=pcr.MemoryLocation.Device))
    pipe.ingest(cloud_from(x, y, {"value": v}))
    pipe.finalize()
    res = pipe.result()
    assert res.location() == pcr.MemoryLocation.Device and res.band_device_ptr(0) != 0
    with pytest.raises(RuntimeError, match="Device memory"):
        res.band_array(0)
    assert_band_close(np.array(res.to_host().band_array(0)), O.run(og, O.COUNT, x, y, v), what="device result")


def test_sharded_pipelines_on_one_gpu_equal_unsharded():
    """Two row-block pipelines (as two ranks would hold) + manual halo merge == one pipeline."""
    import ctypes as C
    from conftest import load_cabi
    A = load_cabi()
    G = 96
    og = O.make_grid((0, 0, G, G))
    rng = np.random.default_rng(9)
    n = 20000
    x, y = rng.uniform(0, G, n), rng.uniform(0, G, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    gs = pcr.gaussian_splat_spec("value", default_sigma=2.0, max_radius_cells=5.0)
    cloud = cloud_from(x, y, {"value": v}, "device")
    shards = []
    for r0, r1 in ((0, 40), (40, 96)):
        p = pcr.Pipeline.create(config_for(og, [gs, spec("Count")], shard_row_begin=r0, shard_row_end=r1))
        p.ingest(cloud)
        p.synchronize()
        shards.append(p)
    top, bot = shards
    halo = top.halo_rows()
    assert halo == 5 and bot.state_row_begin() == 35 and top.state_row_count() == 45
    L = A.lib()
    for (tp, kind, _), (bp, kind2, _) in zip(top.state_planes(), bot.state_planes()):
        assert kind == kind2
        if tp == 0:
            continue
        row = G * 4
        # glyph planes have a halo; the Count (Point) plane shares the window but gets no apron data
        A.check(L.pcr_hip_plane_merge(kind, C.c_void_p(bp + halo * row), C.c_void_p(tp + 40 * row), halo * G, None))
        A.check(L.pcr_hip_plane_merge(kind, C.c_void_p(tp + 35 * row), C.c_void_p(bp), halo * G, None))
    A.check(L.pcr_hip_device_synchronize())
    for p in shards:
        p.finalize()
    got_g = np.vstack([np.array(top.result().band_array(0)), np.array(bot.result().band_array(0))])
    got_c = np.vstack([np.array(top.result().band_array(1)), np.array(bot.result().band_array(1))])
    want_g = O.run(og, O.WEIGHTED_AVERAGE, x, y, v, glyph=O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=2.0, sigma_y=2.0, max_radius=5.0))
    assert_band_close(got_g, want_g, rtol=1e-4, atol=1e-6, what="sharded gaussian")
    assert_band_close(got_c, O.run(og, O.COUNT, x, y, v), what="sharded count")


@pytest.mark.parametrize("result_on", ["host", "device"])
def test_file_in_file_out(tmp_path, result_on):
    """PCRP file -> HBM -> ingest -> finalize -> GeoTIFF at output_path (pipeline.cpp:1351-1361 of the
    reference), read back and compared with the in-memory result and the oracle."""
    og = O.make_grid((0.0, 0.0, 200.0, 120.0), tile=(64, 64))
    rng = np.random.default_rng(21)
    n = 50_000
    x, y = rng.uniform(0, 200, n), rng.uniform(0, 120, n)
    v = rng.normal(5.0, 2.0, n).astype(np.float32)
    src = str(tmp_path / "in.pcrp")
    pcr.write_point_cloud(src, cloud_from(x, y, {"value": v}))
    cloud = pcr.read_point_cloud(src, pcr.PointCloudFormat.Auto, pcr.MemoryLocation.Device)
    assert cloud.location() == pcr.MemoryLocation.Device and cloud.count() == n
    out = str(tmp_path / "out.tif")
    cfg = config_for(og, [spec("Count"), spec("Average")], output_path=out)
    cfg.grid.crs = pcr.CRS.from_epsg(32633)
    if result_on == "device":
        cfg.result_location = pcr.MemoryLocation.Device
    pipe = pcr.Pipeline.create(cfg)
    assert pipe is not None, pcr.pipeline_create_error()
    pipe.ingest(cloud)
    pipe.finalize()
    w, h, nb, crs, bounds = pcr.read_geotiff_info(out)
    assert (w, h, nb) == (200, 120, 2) and crs.epsg == 32633
    assert (bounds.min_x, bounds.min_y, bounds.max_x, bounds.max_y) == (0.0, 0.0, 200.0, 120.0)
    res = pipe.result() if result_on == "host" else pipe.result().to_host()
    for b, name in enumerate(("Count", "Average")):
        got = pcr.read_geotiff_band(out, b)
        assert np.array_equal(got, res.band_array(b), equal_nan=True)
        want = O.run(og, ORT[name], x, y, v)
        if name == "Count":
            assert_band_close(got, want, what="Count from file")
        else:
            assert_band_close(got, want, rtol=2e-5, atol=1e-4, what="Average from file")
    assert pcr.read_geotiff_band_names(out) == [res.band_desc(0).name, res.band_desc(1).name]


def test_ingest_file_streams_chunks_and_ingest_async():
    """Double-buffered file streaming equals one ingest of the whole cloud."""
    import os
    import tempfile
    og = O.make_grid((0.0, 0.0, 256.0, 128.0))
    rng = np.random.default_rng(33)
    n = 300_000
    x, y = rng.uniform(-1, 257, n), rng.uniform(-1, 129, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    whole = cloud_from(x, y, {"value": v})
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "s.pcrp")
        pcr.write_point_cloud(path, whole)
        pipe = pcr.Pipeline.create(config_for(og, [spec("Count"), spec("Max")]))
        assert pipe.ingest_file(path, chunk_points=47_000) == n          # 7 chunks, the last one ragged
        pipe.finalize()
        assert pipe.stats().points_processed == n
        with pytest.raises(RuntimeError, match="failed to open point cloud file"):
            pipe.ingest_file(os.path.join(d, "missing.pcrp"))
    np.testing.assert_array_equal(pipe.result().band_array(0), O.run(og, O.COUNT, x, y, v))
    np.testing.assert_array_equal(pipe.result().band_array(1), O.run(og, O.MAX, x, y, v))
    # ingest_async on page-locked chunks: nothing is waited for until finalize
    pipe2 = pcr.Pipeline.create(config_for(og, [spec("Count")]))
    halves = []
    for sl in (slice(0, n // 2), slice(n // 2, n)):
        c = pcr.PointCloud.create(n, pcr.MemoryLocation.HostPinned)
        c.set_x_array(x[sl]); c.set_y_array(y[sl])
        c.add_channel("value", pcr.DataType.Float32)
        c.set_channel_array_f32("value", v[sl])
        halves.append(c)
        pipe2.ingest_async(c)
    pipe2.finalize()
    np.testing.assert_array_equal(pipe2.result().band_array(0), O.run(og, O.COUNT, x, y, v))
