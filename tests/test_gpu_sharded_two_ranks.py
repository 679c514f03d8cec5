"""Two ranks (two processes, ONE GPU, gloo transport staged through host memory) run the real
ShardedPipeline path: row-block pipelines on the device, zero-copy torch views of the engine's
planes, neighbour halo reduce, touched-tile union, per-rank finalize.  Result must equal the
unsharded CPU oracle.  (RCCL transport itself needs two GPUs: the driver's multi-GPU bench.)"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G_W, G_H, N = 160, 120, 30000


def _inputs():
    rng = np.random.default_rng(21)
    x, y = rng.uniform(0, G_W, N), rng.uniform(0, G_H, N)
    v = rng.uniform(0, 1, N).astype(np.float32)
    return x, y, v


def _worker(rank, world, port, out_dir, tile_h):
    import torch                                   # before pcr: one shared HIP runtime
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    import pcr
    from pcr.distributed import ShardedPipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, y, v = _inputs()
        cfg = pcr.PipelineConfig()
        cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G_W), float(G_H))
        cfg.grid.tile_width, cfg.grid.tile_height = 64, tile_h  # multi-tile: touched flags matter
        cfg.grid.compute_dimensions()
        cfg.exec_mode = pcr.ExecutionMode.GPU
        cnt = pcr.ReductionSpec()
        cnt.value_channel, cnt.type = "value", pcr.ReductionType.Count
        mx = pcr.ReductionSpec()
        mx.value_channel, mx.type = "value", pcr.ReductionType.Max
        cfg.reductions = [pcr.gaussian_splat_spec("value", default_sigma=2.0, max_radius_cells=6.0), cnt, mx]
        sp = ShardedPipeline(cfg, rank, world, device_id=0)
        cloud = pcr.PointCloud.create(N)
        # rank 1 only sees the right half of the cloud's points plus everything in its own rows:
        # the engine filters by centre row, so any superset of a rank's points is fine
        cloud.set_x_array(x)
        cloud.set_y_array(y)
        cloud.add_channel("value", pcr.DataType.Float32)
        cloud.set_channel_array_f32("value", v)
        sp.ingest(cloud.to_device())
        sp.finalize()
        res = sp.result()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), own=np.array(sp.own), halo=sp.halo, local=sp.tiles_local,
                 **{f"b{i}": np.array(res.band_array(i)) for i in range(3)})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tile_h", [64, 60], ids=["blocks-cut-tiles:exchange", "blocks-on-tile-rows:no-collective"])
def test_two_rank_sharded_pipeline_matches_oracle(tmp_path, tile_h):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pcr_oracle_py as O
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), tile_h), nprocs=2, join=True)
    x, y, v = _inputs()
    og = O.make_grid((0, 0, G_W, G_H), tile=(64, tile_h))
    want = [O.run(og, O.WEIGHTED_AVERAGE, x, y, v, glyph=O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=2.0, sigma_y=2.0, max_radius=6.0)),
            O.run(og, O.COUNT, x, y, v), O.run(og, O.MAX, x, y, v)]
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(2)]
    assert parts[0]["own"].tolist() == [0, 60] and parts[1]["own"].tolist() == [60, 120]
    assert int(parts[0]["halo"]) == 6 and bool(parts[0]["local"]) == (tile_h == 60)
    for b, (rt, at) in enumerate([(1e-4, 1e-6), (0, 0), (0, 0)]):
        got = np.vstack([parts[0][f"b{b}"], parts[1][f"b{b}"]])
        w = want[b]
        assert np.array_equal(np.isnan(got), np.isnan(w)), f"band {b}: NaN mask"
        m = ~np.isnan(w)
        assert (np.abs(got[m] - w[m]) <= at + rt * np.abs(w[m])).all(), f"band {b}"


def _point_inputs():
    """Rank 0 (rows [0, 60)) sees points in tile row 0 (rows 0..63, every tile column); rank 1 (rows [60, 120)) only in tile
    (1, 0) -- but it OWNS rows 60..63 of tile row 0, which its own flags leave untouched."""
    rng = np.random.default_rng(5)
    n0, n1 = 20000, 6000
    x = np.concatenate([rng.uniform(0, G_W, n0), rng.uniform(0, 64, n1)])
    y = np.concatenate([rng.uniform(G_H - 56, G_H, n0), rng.uniform(0, G_H - 70, n1)])      # north-up: row = G_H - y
    v = rng.uniform(-1, 1, n0 + n1).astype(np.float32)
    return x, y, v


def _worker_points(rank, world, port, out_dir):
    import torch                                   # before pcr: one shared HIP runtime
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    import pcr
    from pcr.distributed import ShardedPipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, y, v = _point_inputs()
        cfg = pcr.PipelineConfig()
        cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G_W), float(G_H))
        cfg.grid.tile_width, cfg.grid.tile_height = 64, 64
        cfg.grid.compute_dimensions()
        cfg.exec_mode = pcr.ExecutionMode.GPU
        cfg.scatter_path = 2                        # binned: the scatter stores the bands, the exchange must drop them where needed
        specs = []
        for t in (pcr.ReductionType.Sum, pcr.ReductionType.Count, pcr.ReductionType.Max):
            r = pcr.ReductionSpec()
            r.value_channel, r.type = "value", t
            specs.append(r)
        cfg.reductions = specs
        sp = ShardedPipeline(cfg, rank, world, device_id=0)
        assert sp.halo == 0 and not sp.tiles_local
        cloud = pcr.PointCloud.create(len(x))
        cloud.set_x_array(x)
        cloud.set_y_array(y)
        cloud.add_channel("value", pcr.DataType.Float32)
        cloud.set_channel_array_f32("value", v)
        sp.ingest(cloud.to_device())
        stored = sp.pipe.last_scatter()["bands_with_scatter"]
        sp.finalize()
        still = sp.pipe.last_scatter()["bands_with_scatter"]       # the flags-only exchange hands nothing out for writing
        res = sp.result()
        np.savez(os.path.join(out_dir, f"p{rank}.npz"), own=np.array(sp.own), stored=stored, still=still,
                 **{f"b{i}": np.array(res.band_array(i)) for i in range(3)})
    finally:
        dist.destroy_process_group()


def test_two_rank_point_pipelines_exchange_flags_only(tmp_path):
    """Point-only shards exchange nothing but the touched flags -- in a copy, merged back on the device -- and the bands the
    scatter stored survive on the rank whose flags the union does not change (rank 0) and are redone on the other."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pcr_oracle_py as O
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_points, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    x, y, v = _point_inputs()
    og = O.make_grid((0, 0, G_W, G_H), tile=(64, 64))
    want = [O.run(og, O.SUM, x, y, v, wide=True), O.run(og, O.COUNT, x, y, v), O.run(og, O.MAX, x, y, v)]
    parts = [np.load(tmp_path / f"p{r}.npz") for r in range(2)]
    assert parts[0]["own"].tolist() == [0, 60] and parts[1]["own"].tolist() == [60, 120]
    assert int(parts[0]["stored"]) == 1 and int(parts[1]["stored"]) == 1 and int(parts[0]["still"]) == 1
    for b, (rt, at) in enumerate([(1e-5, 1e-6), (0, 0), (0, 0)]):
        got = np.vstack([parts[0][f"b{b}"], parts[1][f"b{b}"]])
        w = want[b]
        assert np.array_equal(np.isnan(got), np.isnan(w)), f"band {b}: NaN mask"
        m = ~np.isnan(w)
        assert (np.abs(got[m] - w[m]) <= at + rt * np.abs(w[m])).all(), f"band {b}"
    # rows 60..63 of rank 1 lie in tile row 0, which only rank 0 touched: Sum there is 0.0, not NaN
    assert not np.isnan(parts[1]["b0"][:4]).any() and (parts[1]["b0"][:4] == 0.0).all()


def _twice_inputs(k):
    rng = np.random.default_rng(77 + k)
    n = 20000
    x, y = rng.uniform(0, G_W, n), rng.uniform(0, G_H, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    d = rng.uniform(0, np.pi, n).astype(np.float32)
    return x, y, v, d


def _twice_cfg(pcr):
    cfg = pcr.PipelineConfig()
    cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G_W), float(G_H))
    cfg.grid.tile_width, cfg.grid.tile_height = 64, 64          # rows [0, 60) / [60, 120) cut tile row 0: every finalize exchanges
    cfg.grid.compute_dimensions()
    cfg.exec_mode = pcr.ExecutionMode.GPU
    line = pcr.line_splat_spec("value", direction_channel="dir", default_half_length=5.0, max_radius_cells=7.0)
    lcount = pcr.line_splat_spec("value", direction_channel="dir", default_half_length=5.0, max_radius_cells=7.0)
    lcount.type = pcr.ReductionType.Count
    gsum = pcr.gaussian_splat_spec("value", default_sigma=1.5, max_radius_cells=5.0)
    gsum.type = pcr.ReductionType.Sum
    cfg.reductions = [pcr.gaussian_splat_spec("value", default_sigma=1.5, max_radius_cells=5.0), gsum, line, lcount]
    return cfg


def _worker_twice(rank, world, port, out_dir, comm):
    import torch                                   # before pcr: one shared HIP runtime
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    import pcr
    from pcr.distributed import ShardedPipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = _twice_cfg(pcr)
        cfg.output_path = os.path.join(out_dir, "whole.tif")        # ONE file, written by rank 0 from the gathered strips
        cfg.result_location = pcr.MemoryLocation.Device if rank == 1 else pcr.MemoryLocation.Host
        sp = ShardedPipeline(cfg, rank, world, device_id=0, comm=comm)
        out = {"own": np.array(sp.own), "halo": sp.halo}
        for k in range(3):                          # ingest -> finalize, three times on ONE pipeline
            x, y, v, d = _twice_inputs(k)
            cloud = pcr.PointCloud.create(len(x))
            cloud.set_x_array(x)
            cloud.set_y_array(y)
            cloud.add_channel("value", pcr.DataType.Float32)
            cloud.set_channel_array_f32("value", v)
            cloud.add_channel("dir", pcr.DataType.Float32)
            cloud.set_channel_array_f32("dir", d)
            sp.ingest(cloud.to_device())
            sp.finalize()
            if k == 1:
                sp.finalize()                       # a finalize with nothing new in between changes nothing either
            res = sp.result()
            if res.location() == pcr.MemoryLocation.Device:
                res = res.to_host()
            for b in range(4):
                out[f"k{k}b{b}"] = np.array(res.band_array(b))
        whole = sp.gather(1)                        # the strips to the LAST rank: one grid
        assert (whole is not None) == (rank == 1)
        if whole is not None:
            for b in range(4):
                out[f"whole{b}"] = np.array(whole.band_array(b))
        np.savez(os.path.join(out_dir, f"t{rank}.npz"), **out)
        sp.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_pipeline_refinalizes_without_counting_the_halo_twice(tmp_path):
    """ingest -> finalize -> ingest -> finalize (-> finalize) -> ingest -> finalize on two row-block shards with a Gaussian and a
    Line group: state survives finalize (src/engine/pipeline.cpp:1344-1364; scripts/benchmarks/benchmark_glyph_full.py:232-246
    re-finalizes one pipeline), so every exchange may carry only what was accumulated since the previous one -- round 4's
    exchange left the apron rows as they were and the owner's boundary rows grew with every finalize."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pcr_oracle_py as O
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_twice, args=(2, port, str(tmp_path), "torch"), nprocs=2, join=True)
    og = O.make_grid((0, 0, G_W, G_H), tile=(64, 64))
    gg = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=1.5, sigma_y=1.5, max_radius=5.0)
    parts = [np.load(tmp_path / f"t{r}.npz") for r in range(2)]
    assert int(parts[0]["halo"]) >= 6
    xs, ys, vs, ds = [], [], [], []
    for k in range(3):
        x, y, v, d = _twice_inputs(k)
        xs.append(x); ys.append(y); vs.append(v); ds.append(d)
        X, Y, V, D = np.concatenate(xs), np.concatenate(ys), np.concatenate(vs), np.concatenate(ds)
        lg = O.make_glyph(O.GLYPH_LINE, half_length=5.0, max_radius=7.0)
        want = [O.run(og, O.WEIGHTED_AVERAGE, X, Y, V, glyph=gg), O.run(og, O.SUM, X, Y, V, glyph=gg),
                O.run(og, O.WEIGHTED_AVERAGE, X, Y, V, glyph=lg, direction=D), O.run(og, O.COUNT, X, Y, V, glyph=lg, direction=D)]
        for b, (rt, at) in enumerate([(1e-4, 1e-6), (1e-4, 1e-5), (1e-4, 1e-6), (0, 0)]):
            got = np.vstack([parts[0][f"k{k}b{b}"], parts[1][f"k{k}b{b}"]])
            w = want[b]
            assert np.array_equal(np.isnan(got), np.isnan(w)), f"finalize {k}, band {b}: NaN mask"
            m = ~np.isnan(w)
            assert (np.abs(got[m] - w[m]) <= at + rt * np.abs(w[m])).all(), f"finalize {k}, band {b}"
    # gather: the strips of the last finalize as ONE grid on rank 1; output_path: ONE GeoTIFF of the whole grid from rank 0
    import pcr
    w, h, nb, _crs, _bounds = pcr.read_geotiff_info(str(tmp_path / "whole.tif"))
    assert (w, h, nb) == (G_W, G_H, 4)
    for b in range(4):
        strips = np.vstack([parts[0][f"k2b{b}"], parts[1][f"k2b{b}"]])
        assert np.array_equal(parts[1][f"whole{b}"], strips, equal_nan=True), f"gather, band {b}"
        assert np.array_equal(np.array(pcr.read_geotiff_band(str(tmp_path / "whole.tif"), b)), strips, equal_nan=True), f"GeoTIFF, band {b}"


def _worker_checkpoint(rank, world, port, out_dir):
    import torch                                   # before pcr: one shared HIP runtime
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    import pcr
    from pcr.distributed import ShardedPipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def cfg_for(tile_h, **kw):
            cfg = _twice_cfg(pcr)
            cfg.grid.tile_height = tile_h
            cfg.grid.compute_dimensions()
            for k, v in kw.items():
                setattr(cfg, k, v)
            return cfg

        def cloud_of(k):
            x, y, v, d = _twice_inputs(k)
            c = pcr.PointCloud.create(len(x))
            c.set_x_array(x)
            c.set_y_array(y)
            c.add_channel("value", pcr.DataType.Float32)
            c.set_channel_array_f32("value", v)
            c.add_channel("dir", pcr.DataType.Float32)
            c.set_channel_array_f32("dir", d)
            return c.to_device()

        ck = os.path.join(out_dir, "ck")
        a = ShardedPipeline(cfg_for(60), rank, world, device_id=0, align=60)      # blocks [0, 60) / [60, 120) = whole tile rows
        assert a.tiles_local
        a.ingest(cloud_of(0))
        a.save_state(ck)
        dist.barrier()
        b = ShardedPipeline(cfg_for(60, state_dir=ck, resume=True), rank, world, device_id=0, align=60)
        b.ingest(cloud_of(1))
        b.finalize()
        out = {f"b{i}": np.array(b.result().band_array(i)) for i in range(4)}
        # blocks that CUT tiles (tile height 64, blocks of 60 rows): a tile has two owners -- the owned rows of every plane are
        # gathered to rank 0, which writes the whole grid's tiles; the shards that resume take their own rows out of them
        ck2 = os.path.join(out_dir, "ck_cut")
        c = ShardedPipeline(cfg_for(64), rank, world, device_id=0)
        assert not c.tiles_local
        c.ingest(cloud_of(0))
        c.save_state(ck2)
        dist.barrier()
        try:
            c.pipe.save_state(os.path.join(out_dir, "never"))          # one rank alone cannot
            out["refused"] = np.array(0)
        except RuntimeError as exc:
            out["refused"] = np.array(int("align = tile_height" in str(exc)))
        d = ShardedPipeline(cfg_for(64, state_dir=ck2, resume=True), rank, world, device_id=0)
        d.ingest(cloud_of(1))
        d.finalize()
        for i in range(4):
            out[f"cut{i}"] = np.array(d.result().band_array(i))
        np.savez(os.path.join(out_dir, f"c{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_two_rank_checkpoint_and_resume(tmp_path):
    """`.pcrt` checkpoints of a sharded pipeline (VERDICT r04 missing 3).  Blocks of whole reference-tile rows: every rank
    writes the tiles it owns, the union is an ordinary checkpoint; resumed by two new shards AND by one unsharded pipeline,
    second cloud ingested, all equal to the oracle over both clouds.  Blocks that cut tiles: the state is gathered to rank 0,
    which writes the tiles; the resuming shards take their own rows out of the files."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    import pcr
    import pcr_oracle_py as O
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_checkpoint, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    parts = [np.load(tmp_path / f"c{r}.npz") for r in range(2)]
    assert all(int(p["refused"]) == 1 for p in parts)
    files = sorted(os.listdir(tmp_path / "ck" / "reduction_0"))
    assert files == [f"tile_{r:04d}_{c:04d}.pcrt" for r in range(2) for c in range(3)]        # 120 / 60 x ceil(160 / 64)
    og = O.make_grid((0, 0, G_W, G_H), tile=(64, 60))
    (x0, y0, v0, d0), (x1, y1, v1, d1) = _twice_inputs(0), _twice_inputs(1)
    X, Y, V, D = np.concatenate([x0, x1]), np.concatenate([y0, y1]), np.concatenate([v0, v1]), np.concatenate([d0, d1])
    gg = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=1.5, sigma_y=1.5, max_radius=5.0)
    lg = O.make_glyph(O.GLYPH_LINE, half_length=5.0, max_radius=7.0)
    want = [O.run(og, O.WEIGHTED_AVERAGE, X, Y, V, glyph=gg), O.run(og, O.SUM, X, Y, V, glyph=gg),
            O.run(og, O.WEIGHTED_AVERAGE, X, Y, V, glyph=lg, direction=D), O.run(og, O.COUNT, X, Y, V, glyph=lg, direction=D)]
    # the same checkpoint in ONE unsharded pipeline
    cfg = _twice_cfg(pcr)
    cfg.grid.tile_height = 60
    cfg.grid.compute_dimensions()
    cfg.state_dir, cfg.resume = str(tmp_path / "ck"), True
    one = pcr.Pipeline.create(cfg)
    c = pcr.PointCloud.create(len(x1))
    c.set_x_array(x1)
    c.set_y_array(y1)
    c.add_channel("value", pcr.DataType.Float32)
    c.set_channel_array_f32("value", v1)
    c.add_channel("dir", pcr.DataType.Float32)
    c.set_channel_array_f32("dir", d1)
    one.ingest(c)
    one.finalize()
    for b, (rt, at) in enumerate([(1e-4, 1e-6), (1e-4, 1e-5), (1e-4, 1e-6), (0, 0)]):
        for name, got in (("two shards", np.vstack([parts[0][f"b{b}"], parts[1][f"b{b}"]])), ("unsharded", np.array(one.result().band_array(b)))):
            w = want[b]
            assert np.array_equal(np.isnan(got), np.isnan(w)), f"{name}, band {b}: NaN mask"
            m = ~np.isnan(w)
            assert (np.abs(got[m] - w[m]) <= at + rt * np.abs(w[m])).all(), f"{name}, band {b}"
    # the cut-tile checkpoint: 2 x 3 tiles of 64 rows written by rank 0 alone, resumed by two shards of 60 rows
    assert sorted(os.listdir(tmp_path / "ck_cut" / "reduction_3")) == [f"tile_{r:04d}_{c:04d}.pcrt" for r in range(2) for c in range(3)]
    og64 = O.make_grid((0, 0, G_W, G_H), tile=(64, 64))
    want64 = [O.run(og64, O.WEIGHTED_AVERAGE, X, Y, V, glyph=gg), O.run(og64, O.SUM, X, Y, V, glyph=gg),
              O.run(og64, O.WEIGHTED_AVERAGE, X, Y, V, glyph=lg, direction=D), O.run(og64, O.COUNT, X, Y, V, glyph=lg, direction=D)]
    for b, (rt, at) in enumerate([(1e-4, 1e-6), (1e-4, 1e-5), (1e-4, 1e-6), (0, 0)]):
        got, w = np.vstack([parts[0][f"cut{b}"], parts[1][f"cut{b}"]]), want64[b]
        assert np.array_equal(np.isnan(got), np.isnan(w)), f"cut tiles, band {b}: NaN mask"
        m = ~np.isnan(w)
        assert (np.abs(got[m] - w[m]) <= at + rt * np.abs(w[m])).all(), f"cut tiles, band {b}"
