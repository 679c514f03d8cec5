"""Multi-GPU routing of an unpartitioned cloud (device-side partition + peer copy).
 (1) pcr_hip_route_count / pcr_hip_route_scatter through the C-ABI against the oracle's world_to_cell;
 (2) two ranks (two processes, one GPU, gloo) each handed an ARBITRARY half of the cloud: ShardedPipeline.ingest_unrouted
     partitions on the device, exchanges the groups (all-to-all) and must reproduce the unsharded oracle -- on the
     down-scaled C5 shape: row blocks that cut the reference tiles, Point/Average + Gaussian sigma=1/Average."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

import pcr_oracle_py as O
from conftest import load_cabi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_route_count_and_scatter_match_oracle_rows():
    A = load_cabi()
    L = A.lib()
    W, H, n = 300, 257, 200_000
    rng = np.random.default_rng(5)
    x = rng.uniform(-5, W + 5, n)
    y = rng.uniform(-5, H + 5, n)
    # edge cases: exactly on the bounds (inclusive), on block boundaries, NaN / inf
    x[:8] = [0.0, W, 150.0, 150.0, np.nan, 10.0, np.inf, 20.0]
    y[:8] = [H, 0.0, H - 100.0, H - 101.0, 5.0, np.nan, 5.0, -np.inf]
    v = rng.normal(0, 1, n).astype(np.float32)
    idx = np.arange(n, dtype=np.int32)
    splits = [0, 100, 100, 190, 257]                      # an empty part in the middle
    nparts = len(splits) - 1
    g = A.make_grid((0, 0, W, H), tile=(128, 64))
    og = O.make_grid((0, 0, W, H), tile=(128, 64))
    want = np.full(n, 255, dtype=np.uint8)
    for i in range(n):
        if i < 8 or i % 50 == 0:                           # the oracle's scalar world_to_cell on a sample + the edge cases
            c, r, ok = O.world_to_cell(og, float(x[i]), float(y[i]))
            want[i] = 255 if not ok else int(np.searchsorted(splits, r, side="right") - 1)
    # vectorised restatement for the rest (inclusive bounds, floor of a true division, clamp)
    inb = (x >= 0) & (x <= W) & (y >= 0) & (y <= H)
    row = np.clip(np.floor((y - H) / -1.0), 0, H - 1)
    owner = np.searchsorted(np.array(splits), row, side="right") - 1
    owner[owner >= nparts] = nparts - 1
    full = np.where(inb, owner, 255).astype(np.uint8)
    chk = (np.arange(n) < 8) | (np.arange(n) % 50 == 0)
    assert np.array_equal(full[chk], want[chk])

    dx, dy = A.DeviceBuffer.from_numpy(x), A.DeviceBuffer.from_numpy(y)
    dv, di = A.DeviceBuffer.from_numpy(v), A.DeviceBuffer.from_numpy(idx)
    ddest, dcount = A.DeviceBuffer(n), A.DeviceBuffer(8 * nparts)
    csplits = (C.c_int32 * (nparts + 1))(*splits)
    A.check(L.pcr_hip_route_count(C.byref(g), csplits, nparts, dx.ptr, dy.ptr, None, n, ddest.ptr, dcount.ptr, None))
    A.check(L.pcr_hip_stream_synchronize(None))
    dest = ddest.to_numpy(np.uint8, (n,))
    counts = dcount.to_numpy(np.uint64, (nparts,))
    assert np.array_equal(dest, full)
    assert counts.tolist() == [int((full == p).sum()) for p in range(nparts)] and counts[1] == 0

    total = int(counts.sum())
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
    dcur = A.DeviceBuffer.from_numpy(starts)
    outs = [A.DeviceBuffer(8 * total), A.DeviceBuffer(8 * total), A.DeviceBuffer(4 * total), A.DeviceBuffer(4 * total)]
    srcs = (C.c_void_p * 4)(dx.ptr.value, dy.ptr.value, dv.ptr.value, di.ptr.value)
    dsts = (C.c_void_p * 4)(*[o.ptr.value for o in outs])
    elem = (C.c_int32 * 4)(8, 8, 4, 4)
    A.check(L.pcr_hip_route_scatter(ddest.ptr, n, nparts, dcur.ptr, 4, srcs, dsts, elem, None))
    A.check(L.pcr_hip_stream_synchronize(None))
    gx, gy = outs[0].to_numpy(np.float64, (total,)), outs[1].to_numpy(np.float64, (total,))
    gv, gi = outs[2].to_numpy(np.float32, (total,)), outs[3].to_numpy(np.int32, (total,))
    assert dcur.to_numpy(np.uint64, (nparts,)).tolist() == (starts + counts).tolist()
    for p in range(nparts):                                 # each group = exactly that owner's points, rows intact
        sl = slice(int(starts[p]), int(starts[p] + counts[p]))
        members = np.sort(gi[sl])
        assert np.array_equal(members, np.nonzero(full == p)[0])
        assert np.array_equal(gx[sl], x[gi[sl]]) and np.array_equal(gy[sl], y[gi[sl]]) and np.array_equal(gv[sl], v[gi[sl]])
    # argument errors are loud
    bad = (C.c_int32 * 3)(0, 200, 100)
    assert L.pcr_hip_route_count(C.byref(g), bad, 2, dx.ptr, dy.ptr, None, n, ddest.ptr, dcount.ptr, None) == 1
    assert b"ascending" in L.pcr_hip_last_error()
    m = C.c_float(-1)
    A.check(L.pcr_hip_absmax_f32(dv.ptr, n, C.byref(m), None))
    assert m.value == np.abs(v).max()


G, TILE, N = 512, 192, 120_000          # 256-row (2 ranks) or ~171-row (3 ranks) blocks; 192-row tiles: block edges cut tile rows


def _inputs():
    rng = np.random.default_rng(77)
    x = rng.uniform(-2, G + 2, N)
    y = rng.uniform(-2, G + 2, N)
    v = rng.uniform(0, 1, N).astype(np.float32)
    w = rng.uniform(1, 2, N).astype(np.float32)           # a second channel travels with the points
    return x, y, v, w


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    import pcr
    from pcr.distributed import ShardedPipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, y, v, w = _inputs()
        cfg = pcr.PipelineConfig()
        cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(G))
        cfg.grid.tile_width, cfg.grid.tile_height = TILE, TILE
        cfg.grid.compute_dimensions()
        cfg.exec_mode = pcr.ExecutionMode.GPU
        avg = pcr.ReductionSpec()
        avg.value_channel, avg.type = "value", pcr.ReductionType.Average
        gs = pcr.gaussian_splat_spec("value", default_sigma=1.0, max_radius_cells=4.0)
        gs.type = pcr.ReductionType.Average
        mx = pcr.ReductionSpec()
        mx.value_channel, mx.type = "other", pcr.ReductionType.Max
        cfg.reductions = [avg, gs, mx]
        sp = ShardedPipeline(cfg, rank, world, device_id=0)
        # an ARBITRARY half of the cloud per rank (interleaved): both halves hold points of both row blocks
        sl = slice(rank, None, world)
        cloud = pcr.PointCloud.create(len(x[sl]))
        cloud.set_x_array(x[sl])
        cloud.set_y_array(y[sl])
        cloud.add_channel("value", pcr.DataType.Float32)
        cloud.set_channel_array_f32("value", v[sl])
        cloud.add_channel("other", pcr.DataType.Float32)
        cloud.set_channel_array_f32("other", w[sl])
        got = sp.ingest_unrouted(cloud)                    # host cloud: moved to the device first
        sp.finalize(timed=True)
        res = sp.result()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), own=np.array(sp.own), halo=sp.halo, received=got,
                 local=sp.tiles_local, coll=np.array(sp.collectives_per_step()), sent=sp.halo_bytes_per_step(),
                 **{f"b{i}": np.array(res.band_array(i)) for i in range(3)})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3], ids=["two-ranks", "three-ranks:inner-rank-with-two-neighbours"])
def test_unrouted_cloud_c5_shape_matches_unsharded_oracle(tmp_path, world):
    import torch.multiprocessing as mp
    from pcr.distributed import row_block
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    x, y, v, w = _inputs()
    og = O.make_grid((0, 0, G, G), tile=(TILE, TILE))
    gl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=1.0, sigma_y=1.0, max_radius=4.0)
    want = [O.run(og, O.AVERAGE, x, y, v), O.run(og, O.AVERAGE, x, y, v, glyph=gl), O.run(og, O.MAX, x, y, w)]
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for r in range(world):
        assert parts[r]["own"].tolist() == list(row_block(r, world, G))
        assert int(parts[r]["halo"]) == 4 and not bool(parts[r]["local"])      # blocks cut tiles: live exchange
        neighbours = (1 if r > 0 else 0) + (1 if r < world - 1 else 0)
        # only the Gaussian group's 2 planes exchange halos (send + recv each): the Point planes' halo rows stay empty
        assert parts[r]["coll"].tolist() == [4 * neighbours, 1]
        assert int(parts[r]["sent"]) == neighbours * 2 * 4 * G * 4               # 2 planes x 4 rows x W x 4 B per edge
    inb = (x >= 0) & (x <= G) & (y >= 0) & (y <= G)
    assert sum(int(p["received"]) for p in parts) == int(inb.sum())             # every valid point reached an owner, once
    for b, (rt, at) in enumerate([(1e-5, 1e-6), (1e-4, 1e-6), (0, 0)]):
        got = np.vstack([p[f"b{b}"] for p in parts])
        wv = want[b]
        assert np.array_equal(np.isnan(got), np.isnan(wv)), f"band {b}: NaN mask"
        m = ~np.isnan(wv)
        assert (np.abs(got[m] - wv[m]) <= at + rt * np.abs(wv[m])).all(), f"band {b}"
