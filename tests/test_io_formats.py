"""File formats either side of the hot path: PCRP / CSV point clouds and GeoTIFF output.
Host-only (no GPU).  Pins:
  * PCRP: a known-answer file built byte by byte from the format specification in the reference header
    (tests/golden/make_pcrp_fixture.py -> spec_cloud.pcrp); the writer must reproduce it exactly, the reader
    must return its contents.  The reference's own expectations (tests/cpp/test_point_cloud_io.cpp) are restated.
  * CSV: exact text for known values (the reference prints with std::setprecision(15)).
  * GeoTIFF: an independent decoder (Pillow/libtiff) must read the same pixels and tags this writer stores;
    the LZW decoder is additionally checked against a Pillow-written file."""
import os
import sys

import numpy as np
import pytest

import pcr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import make_pcrp_fixture as fx  # noqa: E402


def make_cloud(x, y, **channels):
    pc = pcr.PointCloud.create(max(len(x), 1))
    pc.set_x_array(np.asarray(x, dtype=np.float64))
    pc.set_y_array(np.asarray(y, dtype=np.float64))
    for name, arr in channels.items():
        arr = np.asarray(arr)
        dt = {np.dtype("float32"): pcr.DataType.Float32, np.dtype("float64"): pcr.DataType.Float64,
              np.dtype("int32"): pcr.DataType.Int32, np.dtype("uint32"): pcr.DataType.UInt32}[arr.dtype]
        pc.add_channel(name, dt)
        pc.channel_array(name)[:] = arr
    return pc


# ---- PCRP ---------------------------------------------------------------------------------------------
def test_pcrp_fixture_is_current():
    assert open(os.path.join(GOLDEN, "spec_cloud.pcrp"), "rb").read() == fx.build()


def test_pcrp_reader_returns_the_known_answer():
    path = os.path.join(GOLDEN, "spec_cloud.pcrp")
    info = pcr.read_point_cloud_info(path)
    n, x, y, chans = fx.cloud()
    assert info.num_points == n and [c.name for c in info.channels] == [c[0] for c in chans]
    assert [int(c.dtype) for c in info.channels] == [c[1] for c in chans]
    assert info.crs.wkt == fx.WKT and info.crs.is_valid()
    pc = pcr.read_point_cloud(path)                       # Auto: by extension
    assert pc.count() == n and pc.crs().wkt == fx.WKT
    np.testing.assert_array_equal(pc.x_array(), x)        # NaN / inf included, bit for bit
    np.testing.assert_array_equal(pc.y_array(), y)
    for name, _, a in chans:
        got = pc.channel_array(name)
        assert got.dtype == a.dtype
        assert got.tobytes() == a.tobytes()


def test_pcrp_writer_reproduces_the_known_answer(tmp_path):
    n, x, y, chans = fx.cloud()
    pc = make_cloud(x, y, **{name: a for name, _, a in chans})
    crs = pcr.CRS()
    crs.wkt = fx.WKT
    pc.set_crs(crs)
    out = str(tmp_path / "w.pcrp")
    pcr.write_point_cloud(out, pc)                        # default format: PCR_Binary
    assert open(out, "rb").read() == fx.build()


def test_pcrp_reference_test_expectations(tmp_path):
    # tests/cpp/test_point_cloud_io.cpp:35-78 (round trip), :127-148 (info), :233-245 (auto-detect by magic),
    # :388-397 (missing file), :409-421 (corrupt), :423-430 (LAS), :528-543 (100 k points)
    rng = np.random.default_rng(0)
    x, y = rng.uniform(0, 100, 100), rng.uniform(0, 100, 100)
    inten = rng.uniform(0, 1, 100).astype(np.float32)
    p = str(tmp_path / "a.pcrp")
    pcr.write_point_cloud(p, make_cloud(x, y, intensity=inten), pcr.PointCloudFormat.PCR_Binary)
    back = pcr.read_point_cloud(p, pcr.PointCloudFormat.PCR_Binary)
    assert back.count() == 100 and back.has_channel("intensity")
    np.testing.assert_array_equal(back.x_array(), x)
    np.testing.assert_array_equal(back.channel_array_f32("intensity"), inten)
    noext = str(tmp_path / "noext")
    os.rename(p, noext)
    assert pcr.read_point_cloud(noext).count() == 100     # detected by magic
    with pytest.raises(RuntimeError):
        pcr.read_point_cloud(str(tmp_path / "missing.pcrp"))
    with pytest.raises(RuntimeError, match="failed to open"):
        pcr.read_point_cloud_info(str(tmp_path / "missing.pcrp"))
    bad = str(tmp_path / "bad.pcrp")
    open(bad, "wb").write(b"\xde\xad\xbe\xef" + b"\0" * 32)
    with pytest.raises(RuntimeError):
        pcr.read_point_cloud(bad)
    with pytest.raises(RuntimeError, match="invalid magic"):
        pcr.read_point_cloud_info(bad)
    with pytest.raises(RuntimeError, match="not yet implemented"):
        pcr.write_point_cloud(str(tmp_path / "c.las"), back, pcr.PointCloudFormat.LAS)
    with pytest.raises(RuntimeError, match="not yet implemented"):
        pcr.read_point_cloud_info(str(tmp_path / "c.las"), pcr.PointCloudFormat.LAS)
    big = str(tmp_path / "big.pcrp")
    n = 100_000
    pcr.write_point_cloud(big, make_cloud(np.arange(n, dtype=float), np.zeros(n), v=np.ones(n, dtype=np.float32)))
    assert pcr.read_point_cloud_info(big).num_points == n
    assert os.path.getsize(big) == 4 + 4 + 8 + 4 + 4 + (2 + 1 + 1) + n * 20
    empty = str(tmp_path / "empty.pcrp")                  # the reference's (disabled) empty-cloud case
    e = pcr.PointCloud.create(1)
    pcr.write_point_cloud(empty, e)
    assert pcr.read_point_cloud(empty).count() == 0


def test_streaming_reader_returns_the_right_rows(tmp_path):
    # test_point_cloud_io.cpp:274-314 only counts points; here every chunk is compared with the file's rows
    # (the reference's PCRP chunk reader walks the SoA body sequentially and returns wrong rows after chunk 0)
    n = 1000
    rng = np.random.default_rng(1)
    x, y = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    cls = rng.integers(0, 9, n).astype(np.int32)
    p = str(tmp_path / "s.pcrp")
    pcr.write_point_cloud(p, make_cloud(x, y, value=v, cls=cls))
    r = pcr.PointCloudReader.open(p)
    assert r.info().num_points == n and len(r.info().channels) == 2
    chunk = pcr.PointCloud.create(128)
    total = 0
    while not r.eof():
        got = r.read_chunk(chunk, 128)
        assert got == min(128, n - total) and chunk.count() == got
        np.testing.assert_array_equal(chunk.x_array(), x[total:total + got])
        np.testing.assert_array_equal(chunk.y_array(), y[total:total + got])
        np.testing.assert_array_equal(chunk.channel_array("value"), v[total:total + got])
        np.testing.assert_array_equal(chunk.channel_array("cls"), cls[total:total + got])
        total += got
    assert total == n and r.read_chunk(chunk, 128) == 0
    r.rewind()                                            # :356-384
    assert not r.eof() and r.read_chunk(chunk, 100) == 100
    np.testing.assert_array_equal(chunk.x_array(), x[:100])


# ---- CSV ----------------------------------------------------------------------------------------------
def test_csv_text_and_round_trip(tmp_path):
    x = np.array([1.5, -2.25, 123456.789012345])
    y = np.array([0.0, 10.0, 1e-3])
    z = np.array([0.1, 2.0, 3.5], dtype=np.float32)
    k = np.array([7, -8, 9], dtype=np.int32)
    p = str(tmp_path / "c.csv")
    pcr.write_point_cloud(p, make_cloud(x, y, z=z, k=k), pcr.PointCloudFormat.Auto)
    assert open(p).read() == ("x,y,k,z\n"
                              "1.5,0,7,0.100000001490116\n"
                              "-2.25,10,-8,2\n"
                              "123456.789012345,0.001,9,3.5\n")
    info = pcr.read_point_cloud_info(p)
    assert info.num_points == 3 and [c.name for c in info.channels] == ["k", "z"]
    assert all(c.dtype == pcr.DataType.Float64 for c in info.channels)          # as upstream
    back = pcr.read_point_cloud(p)
    assert back.count() == 3
    np.testing.assert_allclose(back.x_array(), x, atol=1e-10)                   # test_point_cloud_io.cpp:199-206
    np.testing.assert_allclose(back.channel_array("z"), z.astype(np.float64), atol=1e-10)
    np.testing.assert_array_equal(back.channel_array("k"), [7.0, -8.0, 9.0])
    with pytest.raises(RuntimeError, match="must start with x,y"):
        bad = str(tmp_path / "bad.csv")
        open(bad, "w").write("a,b\n1,2\n")
        pcr.read_point_cloud_info(bad)
    # streaming, 500 rows in chunks of 50 (:316-354)
    n = 500
    q = str(tmp_path / "s.csv")
    xs = np.arange(n) * 0.5
    pcr.write_point_cloud(q, make_cloud(xs, -xs, w=(xs * 2).astype(np.float64)), pcr.PointCloudFormat.CSV)
    r = pcr.PointCloudReader.open(q)
    assert r.info().num_points == n
    chunk = pcr.PointCloud.create(50)
    total = 0
    while True:
        got = r.read_chunk(chunk, 50)
        if got == 0:
            break
        np.testing.assert_array_equal(chunk.x_array(), xs[total:total + got])
        np.testing.assert_array_equal(chunk.channel_array("w"), xs[total:total + got] * 2)
        total += got
    assert total == n and r.eof()


# ---- GeoTIFF --------------------------------------------------------------------------------------------
def make_grid(W, H, names, seed=0):
    bands = []
    for nm in names:
        b = pcr.BandDesc()
        b.name = nm
        bands.append(b)
    g = pcr.Grid.create(W, H, bands)
    rng = np.random.default_rng(seed)
    data = []
    for i in range(len(names)):
        a = rng.normal(0, 100, (H, W)).astype(np.float32)
        a[rng.uniform(size=(H, W)) < 0.2] = np.nan            # nodata cells
        a[0, 0] = np.inf
        g.set_band_array(i, a)
        data.append(a)
    return g, data


def grid_config(W, H, cell=(2.0, -2.0), origin=(500000.0, 4100000.0), epsg=32618, tile=(4096, 4096)):
    c = pcr.GridConfig()
    c.bounds = pcr.BBox(origin[0], origin[1] - H * abs(cell[1]), origin[0] + W * cell[0], origin[1])
    c.cell_size_x, c.cell_size_y = cell
    c.tile_width, c.tile_height = tile
    c.compute_dimensions()
    assert (c.width, c.height) == (W, H)
    if epsg:
        c.crs = pcr.CRS.from_epsg(epsg)
    return c


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("compress", ["NONE", "LZW", "DEFLATE"])
@pytest.mark.parametrize("tiled", [True, False])
@pytest.mark.parametrize("big", [True, False])
def test_geotiff_round_trip_and_independent_decoder(tmp_path, compress, tiled, big):
    from PIL import Image
    W, H = 301, 203                                     # ragged against 64-pixel tiles and 16-row strips
    g, data = make_grid(W, H, ["elevation mean <m>"])
    cfg = grid_config(W, H)
    opt = pcr.GeoTiffOptions()
    opt.compress, opt.bigtiff = compress, big
    opt.tile_width, opt.tile_height = (64, 64) if tiled else (0, 0)
    p = str(tmp_path / "o.tif")
    pcr.write_geotiff(p, g, cfg, opt)
    # this build's reader
    w, h, nb, crs, bounds = pcr.read_geotiff_info(p)
    assert (w, h, nb) == (W, H, 1) and crs.epsg == 32618
    assert (bounds.min_x, bounds.max_y, bounds.max_x, bounds.min_y) == (500000.0, 4100000.0, 500000.0 + 2 * W, 4100000.0 - 2 * H)
    assert same(pcr.read_geotiff_band(p, 0), data[0])
    assert pcr.read_geotiff_band_names(p) == ["elevation mean <m>"]
    # an independent decoder: Pillow + libtiff
    im = Image.open(p)
    assert im.mode == "F" and im.size == (W, H)
    assert same(np.array(im), data[0])
    tags = im.tag_v2
    assert tuple(tags[33550]) == (2.0, 2.0, 0.0)                               # ModelPixelScale
    assert tuple(tags[33922]) == (0.0, 0.0, 0.0, 500000.0, 4100000.0, 0.0)     # ModelTiepoint
    assert tags[42113].rstrip("\0") == "nan"                                   # GDAL_NODATA
    assert 'role="description">elevation mean &lt;m&gt;</Item>' in tags[42112]
    keys = list(tags[34735])
    assert keys[:4] == [1, 1, 0, 3] and keys[4:8] == [1024, 0, 1, 1] and keys[12:16] == [3072, 0, 1, 32618]
    assert tags[259] == {"NONE": 1, "LZW": 5, "DEFLATE": 8}[compress]


def test_geotiff_multiband_planes(tmp_path):
    W, H = 130, 70
    names = ["sum", "count", "avg"]
    g, data = make_grid(W, H, names, seed=3)
    cfg = grid_config(W, H, epsg=4326, cell=(0.5, -0.5), origin=(-10.0, 50.0))
    p = str(tmp_path / "m.tif")
    pcr.write_geotiff(p, g, cfg)                        # defaults: LZW, 256 x 256 tiles, BigTIFF
    w, h, nb, crs, bounds = pcr.read_geotiff_info(p)
    assert (w, h, nb) == (W, H, 3) and crs.epsg == 4326 and crs.is_geographic()
    for b in range(3):
        assert same(pcr.read_geotiff_band(p, b), data[b])
    assert pcr.read_geotiff_band_names(p) == names
    with pytest.raises(RuntimeError, match="band index out of range"):
        pcr.read_geotiff_band(p, 3)
    # structure as an independent (pure Python) BigTIFF directory walk sees it
    import struct
    raw = open(p, "rb").read()
    assert raw[:4] == b"II+\0" and struct.unpack_from("<HH", raw, 4) == (8, 0)       # BigTIFF header
    (ifd_off,) = struct.unpack_from("<Q", raw, 8)
    (nent,) = struct.unpack_from("<Q", raw, ifd_off)
    tsize = {1: 1, 2: 1, 3: 2, 4: 4, 12: 8, 16: 8}
    fmt = {1: "B", 3: "H", 4: "I", 12: "d", 16: "Q"}
    ifd = {}
    for i in range(nent):
        tag, typ, cnt = struct.unpack_from("<HHQ", raw, ifd_off + 8 + 20 * i)
        voff = ifd_off + 8 + 20 * i + 12
        if tsize[typ] * cnt > 8:
            (voff,) = struct.unpack_from("<Q", raw, voff)
        ifd[tag] = raw[voff:voff + cnt] if typ == 2 else struct.unpack_from(f"<{cnt}{fmt[typ]}", raw, voff)
    assert list(ifd) == sorted(ifd)                                                 # tags ascending
    assert ifd[256] == (W,) and ifd[257] == (H,) and ifd[277] == (3,) and ifd[284] == (2,)
    assert ifd[258] == (32, 32, 32) and ifd[339] == (3, 3, 3) and ifd[259] == (5,)
    assert ifd[322] == (256,) and ifd[323] == (256,) and len(ifd[324]) == 3 == len(ifd[325])   # one tile per plane
    assert ifd[33550] == (0.5, 0.5, 0.0) and ifd[33922] == (0.0, 0.0, 0.0, -10.0, 50.0, 0.0)
    assert ifd[34735][4:8] == (1024, 0, 1, 2) and ifd[34735][12:16] == (2048, 0, 1, 4326)  # geographic, EPSG:4326
    assert ifd[42113] == b"nan\0"
    # unsupported requests fail loudly
    opt = pcr.GeoTiffOptions()
    opt.compress = "ZSTD"
    with pytest.raises(RuntimeError, match="not available"):
        pcr.write_geotiff(p, g, cfg, opt)
    opt = pcr.GeoTiffOptions()
    opt.cloud_optimized = True
    with pytest.raises(RuntimeError, match="not implemented"):
        pcr.write_geotiff(p, g, cfg, opt)
    bad = grid_config(W + 1, H)
    with pytest.raises(RuntimeError, match="grid dimensions mismatch config"):
        pcr.write_geotiff(p, g, bad)


def test_lzw_decoder_reads_a_libtiff_file(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(5)
    a = rng.integers(0, 4, (90, 140)).astype(np.float32)     # compressible: long LZW strings, table resets
    a = np.kron(a, np.ones((4, 4), dtype=np.float32))
    p = str(tmp_path / "pil_lzw.tif")
    Image.fromarray(a, mode="F").save(p, compression="tiff_lzw")
    w, h, nb, _, _ = pcr.read_geotiff_info(p)
    assert (w, h, nb) == (a.shape[1], a.shape[0], 1)
    assert same(pcr.read_geotiff_band(p, 0), a)
    # and the other direction on the same data: this encoder -> libtiff
    b = pcr.BandDesc()
    b.name = "a"
    g = pcr.Grid.create(a.shape[1], a.shape[0], [b])
    g.set_band_array(0, a)
    q = str(tmp_path / "ours_lzw.tif")
    opt = pcr.GeoTiffOptions()
    opt.tile_width = opt.tile_height = 0
    opt.bigtiff = False
    pcr.write_geotiff(q, g, grid_config(a.shape[1], a.shape[0], cell=(1.0, -1.0), origin=(0.0, float(a.shape[0])), epsg=0), opt)
    assert same(np.array(Image.open(q)), a)
    assert os.path.getsize(q) < a.nbytes // 8               # it does compress


@pytest.mark.parametrize("tile,tiff_tile", [((128, 64), (64, 32)), ((100, 60), (64, 64)), ((128, 64), (0, 0))],
                         ids=["aligned-stream", "unaligned-buffered", "strips-buffered"])
def test_tiled_writer_assembles_reference_tiles(tmp_path, tile, tiff_tile):
    from PIL import Image
    W, H = 300, 150
    cfg = grid_config(W, H, tile=tile)
    rng = np.random.default_rng(9)
    full = rng.normal(size=(2, H, W)).astype(np.float32)
    opt = pcr.GeoTiffOptions()
    opt.tile_width, opt.tile_height = tiff_tile
    opt.compress = "DEFLATE"
    p = str(tmp_path / "t.tif")
    w = pcr.TiledGeoTiffWriter.open(p, cfg, ["a", "b"], opt)
    skipped = (1, 2)
    order = [(r, c) for r in range(cfg.tiles_y) for c in range(cfg.tiles_x)]
    for r, c in reversed(order):                        # any order
        if (r, c) == skipped:
            continue
        c0, r0, cols, rows = cfg.tile_cell_range(pcr.TileIndex(r, c))
        w.write_tile(r, c, full[:, r0:r0 + rows, c0:c0 + cols].copy())
    with pytest.raises(RuntimeError, match="band count mismatch"):
        w.write_tile(0, 0, full[:1, :tile[1], :tile[0]].copy())
    w.close()
    want = full.copy()
    c0, r0, cols, rows = cfg.tile_cell_range(pcr.TileIndex(*skipped))
    want[:, r0:r0 + rows, c0:c0 + cols] = np.nan         # never written -> nodata
    for b in range(2):
        assert same(pcr.read_geotiff_band(p, b), want[b])
    assert pcr.read_geotiff_band_names(p) == ["a", "b"]
    assert Image.open(p).size == (W, H)                 # libtiff accepts the directory
