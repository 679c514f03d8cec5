"""The N > 1 path on CPU: row-block partition + neighbour halo reduce over gloo (world_size 2 and 3).

Each rank's state planes are produced by the CPU oracle restricted to the rank's row window
(own rows +- halo), exactly what a sharded device engine holds; after exchange_halos the owned
rows must equal the unsharded oracle result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    import pcr_oracle_py as O
    from pcr.distributed import exchange_halos, row_block, allreduce_touched, PLANE_SUM, PLANE_WGT, PLANE_MAX
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H, halo = 48, 60, 6
        og = O.make_grid((0, 0, W, H))
        rng = np.random.default_rng(3)
        n = 4000
        x, y = rng.uniform(0, W, n), rng.uniform(0, H, n)
        v = rng.uniform(0, 1, n).astype(np.float32)
        gl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=2.0, sigma_y=2.0, max_radius=float(halo))
        blocks = [row_block(r, world, H) for r in range(world)]
        r0, r1 = blocks[rank]
        s0, s1 = max(0, r0 - halo), min(H, r1 + halo)
        # what this rank's engine would hold: only points whose centre row is in [r0, r1),
        # footprints accumulated over rows [s0, s1)
        rows = np.floor((y - og.max_y) / og.cell_size_y).clip(0, H - 1).astype(int)
        mine = (rows >= r0) & (rows < r1)
        num = O.run(og, O.SUM, x[mine], y[mine], v[mine], glyph=gl) if mine.any() else np.zeros((H, W), np.float32)
        den = O.run(og, O.COUNT, x[mine], y[mine], v[mine], glyph=gl) if mine.any() else np.full((H, W), np.nan, np.float32)
        num = np.nan_to_num(num)[s0:s1].copy()
        den = np.nan_to_num(den)[s0:s1].copy()
        mx = np.full((s1 - s0, W), -3.4e38, np.float32)
        mx[(r0 - s0):(r1 - s0)] = rank            # a max plane: owned rows carry the rank id
        if s0 < r0:
            mx[: r0 - s0] = 100 + rank            # apron rows destined for the upper neighbour
        planes = [(torch.from_numpy(num), PLANE_SUM), (torch.from_numpy(den), PLANE_WGT),
                  (torch.from_numpy(mx), PLANE_MAX)]
        exchange_halos(planes, (r0, r1), s0, halo, rank, world, blocks=blocks)
        # the apron rows that went to their owners hold the identity again
        if rank > 0:
            assert (num[: r0 - s0] == 0).all() and (den[: r0 - s0] == 0).all() and (mx[: r0 - s0] < -3e38).all()
        if rank < world - 1:
            assert (num[r1 - s0:] == 0).all() and (den[r1 - s0:] == 0).all()
        first = dict(num=num[(r0 - s0):(r1 - s0)].copy(), den=den[(r0 - s0):(r1 - s0)].copy())
        # ingest -> finalize -> ingest -> finalize (state survives finalize, src/engine/pipeline.cpp:1344-1364): a second
        # cloud lands on the same planes (the tensors view these arrays) and a second exchange must carry only that
        x2, y2 = rng.uniform(0, W, n), rng.uniform(0, H, n)
        v2 = rng.uniform(0, 1, n).astype(np.float32)
        rows2 = np.floor((y2 - og.max_y) / og.cell_size_y).clip(0, H - 1).astype(int)
        m2 = (rows2 >= r0) & (rows2 < r1)
        if m2.any():
            num += np.nan_to_num(O.run(og, O.SUM, x2[m2], y2[m2], v2[m2], glyph=gl))[s0:s1]
            den += np.nan_to_num(O.run(og, O.COUNT, x2[m2], y2[m2], v2[m2], glyph=gl))[s0:s1]
        exchange_halos(planes, (r0, r1), s0, halo, rank, world, blocks=blocks)
        np.savez(os.path.join(out_dir, f"second{rank}.npz"), num=num[(r0 - s0):(r1 - s0)], den=den[(r0 - s0):(r1 - s0)])
        num2, den2 = num, den
        num, den = first["num"], first["den"]
        mx_own = mx[(r0 - s0):(r1 - s0)]
        touched = torch.tensor([1 if rank == 1 else 0, 0], dtype=torch.int32)
        allreduce_touched(touched)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), num=num, den=den,
                 mx=mx_own, own=np.array([r0, r1]), touched=touched.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_matches_unsharded(world, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pcr_oracle_py as O
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    W, H, halo = 48, 60, 6
    og = O.make_grid((0, 0, W, H))
    rng = np.random.default_rng(3)
    n = 4000
    x, y = rng.uniform(0, W, n), rng.uniform(0, H, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    gl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=2.0, sigma_y=2.0, max_radius=float(halo))
    num = np.nan_to_num(O.run(og, O.SUM, x, y, v, glyph=gl))
    den = np.nan_to_num(O.run(og, O.COUNT, x, y, v, glyph=gl))
    x2, y2 = rng.uniform(0, W, n), rng.uniform(0, H, n)
    v2 = rng.uniform(0, 1, n).astype(np.float32)
    num2 = num + np.nan_to_num(O.run(og, O.SUM, x2, y2, v2, glyph=gl))
    den2 = den + np.nan_to_num(O.run(og, O.COUNT, x2, y2, v2, glyph=gl))
    rows_seen = 0
    for r in range(world):
        d = np.load(tmp_path / f"r{r}.npz")
        r0, r1 = d["own"]
        rows_seen += r1 - r0
        np.testing.assert_allclose(d["num"], num[r0:r1], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(d["den"], den[r0:r1], rtol=1e-5, atol=1e-6)
        # after the SECOND exchange: both clouds, every contribution exactly once (round 4 counted the first halo twice)
        d2 = np.load(tmp_path / f"second{r}.npz")
        np.testing.assert_allclose(d2["num"], num2[r0:r1], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(d2["den"], den2[r0:r1], rtol=1e-5, atol=1e-5)
        # max plane: my last `halo` rows received the lower neighbour's apron (100 + its rank)
        if r < world - 1:
            assert (d["mx"][-halo:] == 100 + r + 1).all() and (d["mx"][:-halo] == r).all()
        else:
            assert (d["mx"] == r).all()
        assert d["touched"].tolist() == [1, 0]
    assert rows_seen == H


def test_row_block_partition():
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    from pcr.distributed import row_block
    assert [row_block(r, 8, 16384) for r in range(8)] == [(2048 * r, 2048 * (r + 1)) for r in range(8)]
    blocks = [row_block(r, 3, 100) for r in range(3)]
    assert blocks[0][0] == 0 and blocks[-1][1] == 100
    assert all(blocks[i][1] == blocks[i + 1][0] for i in range(2))
    # tile-aligned blocks (exchange-free for tile-clipped glyphs): 10 tiles of 4096 over 4 ranks
    blocks = [row_block(r, 4, 40000, align=4096) for r in range(4)]
    assert all(b0 % 4096 == 0 for b0, _ in blocks) and blocks[-1][1] == 40000
    assert row_block(0, 1, 77) == (0, 77)


class _FakePipe:
    """Stands in for the device pipeline: what ShardedPipeline.ingest asks of it before anything is accumulated."""

    def __init__(self, need):
        self.need = need
        self.ingested = 0

    def line_reach_rows(self, cloud):
        return self.need

    def ingest(self, cloud):
        self.ingested += 1


def _reach_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    from pcr.distributed import ShardedPipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        log = []
        # round 1: only rank 1 holds a long segment -> BOTH ranks must refuse, neither may have accumulated;
        # round 2: every need fits the halo -> both ingest
        for needs in ([3, 20], [3, 8]):
            sp = ShardedPipeline.__new__(ShardedPipeline)
            sp.world, sp.rank, sp.group, sp.halo = world, rank, dist.group.WORLD, 8
            sp._line_hl_groups, sp.tiles_local = True, False
            sp.pipe = _FakePipe(needs[rank])
            sp._comm = None
            try:
                sp.ingest(object())
                log.append(("ok", sp.pipe.ingested))
            except RuntimeError as e:
                log.append(("refused", sp.pipe.ingested, "reaches 20 rows" in str(e)))
            sp.pipe = None                      # nothing for close() to release
        dist.barrier()                          # a rank that raised alone would have left the other one here
        with open(os.path.join(out_dir, f"reach{rank}.txt"), "w") as f:
            f.write(repr(log))
    finally:
        dist.destroy_process_group()


def test_line_reach_is_agreed_before_anything_is_accumulated(tmp_path):
    """ShardedPipeline.ingest: the largest Line reach over all ranks decides, so that every rank refuses the round
    (or none does) -- checked over gloo with a stand-in for the device pipeline."""
    mp.spawn(_reach_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        log = eval((tmp_path / f"reach{r}.txt").read_text())
        assert log == [("refused", 0, True), ("ok", 1)], (r, log)


class _StripPipe:
    """Stands in for a finalized device pipeline: result() is this rank's strip as a host pcr.Grid."""

    def __init__(self, grid):
        self._grid = grid

    def result(self):
        return self._grid


def _gather_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    import pcr
    import pcr_oracle_py as O
    from pcr.distributed import ShardedPipeline, row_block
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H = 48, 100
        og = O.make_grid((0, 0, W, H))
        rng = np.random.default_rng(8)
        x, y = rng.uniform(0, W, 3000), rng.uniform(0, H, 3000)
        v = rng.uniform(0, 1, 3000).astype(np.float32)
        bands = [O.run(og, O.SUM, x, y, v), O.run(og, O.COUNT, x, y, v)]
        sp = ShardedPipeline.__new__(ShardedPipeline)
        sp.world, sp.rank, sp.group, sp._comm = world, rank, dist.group.WORLD, None
        sp.blocks = [row_block(r, world, H) for r in range(world)]
        sp.own = sp.blocks[rank]
        sp.width = W
        gc = pcr.GridConfig()
        gc.bounds = pcr.BBox(0.0, 0.0, float(W), float(H))
        gc.compute_dimensions()
        sp.grid = gc
        descs = []
        for name in ("value_0", "value_5"):
            d = pcr.BandDesc()
            d.name = name
            descs.append(d)
        r0, r1 = sp.own
        strip = pcr.Grid.create(W, r1 - r0, descs)
        for b in range(2):
            strip.set_band_array(b, bands[b][r0:r1])
        sp.pipe = _StripPipe(strip)
        for dst in (0, world - 1):
            whole = sp.gather(dst)
            assert (whole is not None) == (rank == dst)
            if whole is not None:
                assert (whole.cols(), whole.rows(), whole.num_bands()) == (W, H, 2) and whole.band_desc(1).name == "value_5"
                np.savez(os.path.join(out_dir, f"gather{dst}.npz"), b0=np.array(whole.band_array(0)), b1=np.array(whole.band_array(1)),
                         w0=bands[0], w1=bands[1])
        sp.pipe = None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_assembles_the_strips_into_one_grid(world, tmp_path):
    """ShardedPipeline.gather over gloo: the ranks' strips (34 / 33 / 33 rows at world 3) land in row order on the chosen rank,
    bit for bit the unsharded band (the reference's result() is one grid: src/engine/pipeline.cpp:1175-1186)."""
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for dst in (0, world - 1):
        d = np.load(tmp_path / f"gather{dst}.npz")
        assert np.array_equal(d["b0"], d["w0"], equal_nan=True) and np.array_equal(d["b1"], d["w1"], equal_nan=True)
