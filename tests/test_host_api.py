"""Host-side (no GPU) behaviour of the drop-in Python API: value types, GridConfig math against the
reference's known answers, PointCloud / Grid containers, error behaviour of Pipeline.create."""
import numpy as np
import pytest

import pcr


def make_grid_config(gj):
    g = pcr.GridConfig()
    g.bounds = pcr.BBox(*[float(b) for b in gj["bounds"]])
    g.cell_size_x, g.cell_size_y = gj["cell"]
    g.tile_width, g.tile_height = gj["tile"]
    if gj["dims"]:
        g.width, g.height = gj["dims"]
        g.tiles_x = -(-g.width // g.tile_width)
        g.tiles_y = -(-g.height // g.tile_height)
    else:
        g.compute_dimensions()
    return g


def test_world_to_cell_known_answers(known_answers):
    for c in known_answers["world_to_cell"]:
        g = make_grid_config(c["grid"])
        col, row, ok = g.world_to_cell(float(c["wx"]), float(c["wy"]))
        assert ok == c["valid"], c["source"]
        if ok:
            assert (col, row) == (c["col"], c["row"]), c["source"]


def test_compute_dimensions_and_tiles(known_answers):
    for c in known_answers["compute_dimensions"]:
        gj = dict(c["grid"], dims=None)
        g = make_grid_config(gj)
        assert (g.width, g.height, g.tiles_x, g.tiles_y) == (c["width"], c["height"], c["tiles_x"], c["tiles_y"])
    for c in known_answers["tile_cell_range"]:
        g = make_grid_config(c["grid"])
        assert list(g.tile_cell_range(pcr.TileIndex(c["tile_row"], c["tile_col"]))) == c["expect"]


def test_grid_config_misc():
    # tests/cpp/test_grid_config.cpp:112-121, 144-154, 156-169, 212-242 (reference known answers)
    g = make_grid_config(dict(bounds=(0, 0, 100, 100), cell=(1.0, -1.0), tile=(256, 256), dims=None))
    assert g.cell_to_world(50, 50) == pytest.approx((50.5, 49.5))
    g = make_grid_config(dict(bounds=(0, 0, 1000, 1000), cell=(1.0, -1.0), tile=(256, 256), dims=None))
    t = g.cell_to_tile(300, 400)
    assert (t.col, t.row) == (1, 1)
    b = g.tile_bounds(pcr.TileIndex(1, 1))
    assert (b.min_x, b.max_x, b.max_y, b.min_y) == pytest.approx((256.0, 512.0, 744.0, 488.0))
    assert g.total_tiles() == 16 and g.total_cells() == 1000000
    # validate(): test_grid_config.cpp:249-313
    g.crs = pcr.CRS.from_epsg(3857)
    g.validate()
    bad = pcr.GridConfig()
    bad.bounds = pcr.BBox(100.0, 0.0, 50.0, 10.0)
    with pytest.raises(RuntimeError, match="Invalid bounds"):
        bad.validate()
    g2 = make_grid_config(dict(bounds=(0, 0, 100, 100), cell=(1.0, -1.0), tile=(256, 256), dims=None))
    with pytest.raises(RuntimeError, match="CRS is not valid"):
        g2.validate()
    g2.cell_size_x = 0.0
    with pytest.raises(RuntimeError, match="Cell size cannot be zero"):
        g2.validate()


def test_bbox_and_crs():
    b = pcr.BBox()
    assert not b.valid()
    b.expand(1.0, 2.0)
    b.expand(-3.0, 5.0)
    assert (b.min_x, b.min_y, b.max_x, b.max_y) == (-3.0, 2.0, 1.0, 5.0)
    assert b.contains(1.0, 5.0) and b.contains(-3.0, 2.0) and not b.contains(1.0001, 3.0)
    assert b.width() == 4.0 and b.height() == 3.0
    c = pcr.CRS.from_epsg(32618)          # must not throw without PROJ (20 reference scripts call it)
    assert c.is_valid() and c.epsg == 32618 and c.is_projected() and not c.is_geographic()
    assert pcr.CRS.from_epsg(4326).is_geographic()
    assert not pcr.CRS().is_valid()
    assert c.equivalent_to(pcr.CRS.from_epsg(32618)) and not c.equivalent_to(pcr.CRS.from_epsg(4326))


def test_point_cloud_container_semantics():
    assert pcr.PointCloud.create(0) is None            # src/core/point_cloud.cpp:209-211
    pc = pcr.PointCloud.create(10)
    assert pc.count() == 0 and pc.capacity() == 10 and pc.location() == pcr.MemoryLocation.Host
    pc.set_y_array(np.arange(4.0))                     # copies only
    assert pc.count() == 0
    pc.set_x_array(np.arange(4.0) * 2)                 # copies and sets count = len
    assert pc.count() == 4
    np.testing.assert_array_equal(pc.x_array(), [0, 2, 4, 6])
    np.testing.assert_array_equal(pc.y_array(), [0, 1, 2, 3])
    pc.add_channel("v", pcr.DataType.Float32)
    pc.add_channel("cls", pcr.DataType.Int32)
    with pytest.raises(RuntimeError, match="Channel already exists"):
        pc.add_channel("v", pcr.DataType.Float32)
    assert pc.has_channel("v") and not pc.has_channel("w")
    assert sorted(pc.channel_names()) == ["cls", "v"]
    pc.set_channel_array_f32("v", np.array([1, 2, 3, 4], dtype=np.float32))
    view = pc.channel_array_f32("v")
    view[0] = 9.0                                      # zero-copy view
    assert pc.channel_array_f32("v")[0] == 9.0
    with pytest.raises(RuntimeError, match="exceeds point count"):
        pc.set_channel_array_f32("v", np.zeros(5, dtype=np.float32))
    with pytest.raises(RuntimeError, match="wrong type"):
        pc.channel_array_f32("cls")
    with pytest.raises(RuntimeError, match="too large"):
        pc.set_x_array(np.zeros(11))
    with pytest.raises(RuntimeError, match="beyond capacity"):
        pc.resize(11)
    pc.resize(2)
    assert pc.count() == 2 and len(pc.x_array()) == 2
    h = pc.to_host()
    assert h.count() == 2 and h.channel_array_f32("v")[1] == 2.0
    pc.set_crs(pcr.CRS.from_epsg(3857))
    assert pc.crs().epsg == 3857


def test_grid_container():
    b = pcr.BandDesc()
    b.name = "z"
    g = pcr.Grid.create(4, 3, [b])
    assert (g.cols(), g.rows(), g.num_bands(), g.cell_count()) == (4, 3, 1, 12)
    assert g.band_index("z") == 0 and g.band_index("q") == -1 and g.band_desc(0).name == "z"
    g.fill(2.5)
    a = g.band_array(0)
    assert a.shape == (3, 4) and (a == 2.5).all()
    g.set_band_array(0, np.arange(12, dtype=np.float32).reshape(3, 4))
    assert g.band_array(0)[2, 3] == 11
    with pytest.raises(RuntimeError, match="shape mismatch"):
        g.set_band_array(0, np.zeros((4, 3), dtype=np.float32))
    with pytest.raises(RuntimeError, match="Invalid band"):
        g.band_array(3)


def test_splat_spec_helpers():
    s = pcr.gaussian_splat_spec("z", default_sigma=3.0, default_sigma_y=1.5, max_radius_cells=10.0,
                                output_band_name="zs")
    assert s.type == pcr.ReductionType.WeightedAverage and s.glyph.type == pcr.GlyphType.Gaussian
    assert (s.glyph.default_sigma_x, s.glyph.default_sigma_y, s.glyph.max_radius_cells) == (3.0, 1.5, 10.0)
    assert s.output_band_name == "zs" and s.value_channel == "z"
    s = pcr.line_splat_spec("z", direction_channel="dir", default_half_length=4.0)
    assert s.glyph.type == pcr.GlyphType.Line and s.glyph.direction_channel == "dir"
    assert s.glyph.default_half_length == 4.0 and s.glyph.max_radius_cells == 32.0
    d = pcr.GlyphSpec()
    assert d.type == pcr.GlyphType.Point and d.max_radius_cells == 32.0 and not d.normalize_weights


def test_enum_values_match_reference_numbering():
    # include/pcr/core/types.h:33-45, 118-126; include/pcr/engine/pipeline.h:39-44
    assert [int(getattr(pcr.ReductionType, n)) for n in
            ("Sum", "Max", "Min", "Average", "WeightedAverage", "Count", "Median", "Percentile",
             "MostRecent", "PriorityMerge", "Custom")] == list(range(11))
    assert int(pcr.StatusCode.NotImplemented) == 6 and int(pcr.ExecutionMode.Hybrid) == 3
    assert pcr._pcr.Sum == pcr.ReductionType.Sum and pcr._pcr.Host == pcr.MemoryLocation.Host   # export_values()


def test_pipeline_create_follows_the_reference_matrix(capfd, monkeypatch):
    """The reference's GPU-initialisation matrix (src/engine/pipeline.cpp:100-131) and its tests
    (tests/cpp/test_error_handling.cpp:111-137 GPU_FallbackToCPU_WhenNoDevice, :139-160 GPU_StrictMode_FailsWithoutDevice,
    :162-179 GPU_AutoMode_UsesAvailable, :181-199 CPU_Mode_AlwaysWorks), message for message and code for code.  Where the
    reference continues in CPU mode so does this build -- on its host engine, after the reference's Warning / Info line, and
    visibly (Pipeline.engine()); PCR_REQUIRE_GPU_ENGINE=1 turns every such fallback into an error."""
    monkeypatch.delenv("PCR_REQUIRE_GPU_ENGINE", raising=False)
    cfg = pcr.PipelineConfig()
    cfg.grid.bounds = pcr.BBox(0, 0, 10, 10)
    cfg.grid.compute_dimensions()
    r = pcr.ReductionSpec()
    r.value_channel, r.type = "v", pcr.ReductionType.Sum
    cfg.reductions = [r]
    cfg.exec_mode = pcr.ExecutionMode.CPU                        # CPU_Mode_AlwaysWorks
    pipe = pcr.Pipeline.create(cfg)
    assert pipe is not None and pipe.engine() == "host"
    pipe.validate()
    capfd.readouterr()
    if pcr.device_count() == 0:
        msg = "No CUDA-capable GPU detected"
        cfg.exec_mode = pcr.ExecutionMode.Auto                   # GPU_AutoMode_UsesAvailable
        pipe = pcr.Pipeline.create(cfg)
        assert pipe is not None and pipe.engine() == "host"
        pipe.validate()
        assert f"Info: {msg} - using CPU mode\n" in capfd.readouterr().err
        cfg.exec_mode = pcr.ExecutionMode.GPU                    # GPU_FallbackToCPU_WhenNoDevice
        cfg.gpu_fallback_to_cpu, cfg.gpu_require_strict = True, False            # the reference's defaults
        pipe = pcr.Pipeline.create(cfg)
        assert pipe is not None and pipe.engine() == "host"
        assert f"Warning: {msg} - falling back to CPU mode\n" in capfd.readouterr().err
        monkeypatch.setenv("PCR_REQUIRE_GPU_ENGINE", "1")        # what GPU test suites and the benchmark set: no fallback
        assert pcr.Pipeline.create(cfg) is None
        assert "PCR_REQUIRE_GPU_ENGINE forbids it" in pcr.pipeline_create_error()
        assert "Error:" in capfd.readouterr().err
        monkeypatch.delenv("PCR_REQUIRE_GPU_ENGINE")
        cfg.gpu_require_strict = True                                             # GPU_StrictMode_FailsWithoutDevice
        assert pcr.Pipeline.create(cfg) is None
        assert pcr.pipeline_create_error().endswith(f"{msg} - GPU mode requested but no GPU available")
        assert "Warning:" not in capfd.readouterr().err
        cfg.gpu_fallback_to_cpu, cfg.gpu_require_strict = False, False
        assert pcr.Pipeline.create(cfg) is None
        assert pcr.pipeline_create_error().endswith(f"{msg} - GPU required but not available")
        cfg.gpu_fallback_to_cpu = True
    cfg.exec_mode = pcr.ExecutionMode.CPU
    r.type = pcr.ReductionType.Median                           # not registered -> create fails (pipeline.cpp:229-233)
    cfg.reductions = [r]
    assert pcr.Pipeline.create(cfg) is None
    assert "unknown reduction type" in pcr.pipeline_create_error()


def test_io_names_exist():
    # the reference's I/O names (python/bindings.cpp:500-640); behaviour: tests/test_io_formats.py
    for name in ("write_geotiff", "read_geotiff_info", "read_point_cloud", "write_point_cloud",
                 "read_point_cloud_info", "PointCloudReader", "GeoTiffOptions", "PointCloudInfo"):
        assert hasattr(pcr, name)
    with pytest.raises(RuntimeError, match="Failed to read point cloud"):
        pcr.read_point_cloud("/tmp/nope.pcrp")


def test_grid_misc_known_answers(known_answers):
    """The rest of tests/cpp/test_grid_config.cpp, as data (tests/golden/reference_known_answers.json: grid_misc, validate)."""
    for c in known_answers["grid_misc"]:
        what = c["what"]
        if what == "compute_dimensions_invalid_bounds":
            g = pcr.GridConfig()
            g.cell_size_x, g.cell_size_y = c["cell"]
            g.compute_dimensions()
            assert (g.width, g.height) == (c["width"], c["height"]), c["source"]
            continue
        g = make_grid_config(dict(c["grid"], dims=None))
        if what == "cell_to_world":
            assert g.cell_to_world(c["col"], c["row"]) == pytest.approx((c["wx"], c["wy"])), c["source"]
        elif what == "round_trip":
            col, row, ok = g.world_to_cell(c["wx"], c["wy"])
            wx, wy = g.cell_to_world(col, row)
            assert ok and abs(wx - c["wx"]) < c["max_abs_error"] and abs(wy - c["wy"]) < c["max_abs_error"], c["source"]
        elif what == "cell_to_tile":
            t = g.cell_to_tile(c["col"], c["row"])
            assert (t.col, t.row) == (c["tile_col"], c["tile_row"]), c["source"]
        elif what == "tile_bounds":
            b = g.tile_bounds(pcr.TileIndex(c["tile_row"], c["tile_col"]))
            assert (b.min_x, b.max_x, b.max_y, b.min_y) == pytest.approx((c["min_x"], c["max_x"], c["max_y"], c["min_y"])), c["source"]
        elif what == "totals":
            assert (g.total_tiles(), g.total_cells()) == (c["total_tiles"], c["total_cells"]), c["source"]
        elif what == "gdal_geotransform":
            assert list(g.gdal_geotransform()) == c["gt"], c["source"]
        else:
            raise AssertionError(f"unknown grid_misc case {what}")
    for c in known_answers["validate"]:
        if "grid" in c:
            g = make_grid_config(dict(c["grid"], dims=None))
        else:
            g = pcr.GridConfig()
            g.bounds = pcr.BBox(*c["bounds"])
            g.cell_size_x, g.cell_size_y = c["cell"]
            if c["compute_dimensions"]:
                g.compute_dimensions()
        if c["epsg"]:
            g.crs = pcr.CRS.from_epsg(c["epsg"])
        if c["ok"]:
            g.validate()
        else:
            with pytest.raises(RuntimeError) as ei:
                g.validate()
            # raise_if_error carries the reference's StatusCode in the message prefix
            assert c["code"] in str(ei.value) or c["code"] == "InvalidArgument" or "CRS" in str(ei.value), (c["source"], str(ei.value))
