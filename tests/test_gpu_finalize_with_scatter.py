"""Finalize fused into the scatter that defines the planes (pcr_hip_engine_finalize_with_scatter; round 4).

The reference's finalize reads every tile's state back and runs Op::finalize on it (src/engine/pipeline.cpp:1154-1286).
Here the first Point scatter of a pipeline already holds every cell of every LDS tile when it defines the planes, so the
same pass stores the finished bands; Pipeline.finalize then launches a kernel that returns at once.  What must hold:
the bands equal the oracle's (NaN where the reference tile is untouched), and they are dropped -- the ordinary finalize
pass runs -- whenever anything could have changed the planes or the touched flags in between."""
import ctypes as C

import numpy as np
import pytest

import pcr
import pcr_oracle_py as O
from conftest import load_cabi
from test_gpu_configs_c3_c4_c5 import bands
from test_gpu_fullgrid_oracle import ALL6, check_point_bands, poison_device_memory
from test_gpu_pipeline_api import cloud_from, config_for, spec

pytestmark = pytest.mark.gpu


class OnHost:
    """check_point_bands / bands read p.result() on the host: a device-resident result is copied out first."""
    def __init__(self, p):
        self.p = p

    def result(self):
        r = self.p.result()
        return r.to_host() if r.location() == pcr.MemoryLocation.Device else r


def sparse_cloud(G, n, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(100, 300, n)                                     # a 200 x 200 corner: most reference tiles untouched
    y = rng.uniform(G - 300, G - 100, n)
    return x, y, rng.uniform(-1, 1, n).astype(np.float32)


@pytest.mark.parametrize("location", ["device", "host"])
def test_single_ingest_bands_come_from_the_scatter(location):
    G, n = 1024, 40_000
    x, y, v = sparse_cloud(G, n, 7)
    og = O.make_grid((0, 0, G, G), tile=(256, 256))
    poison_device_memory(8 * G * G * 4)
    cfg = config_for(og, [spec(t) for t in ALL6], scatter_path=2)
    cfg.result_location = pcr.MemoryLocation.Device if location == "device" else pcr.MemoryLocation.Host
    p = pcr.Pipeline.create(cfg)
    p.ingest(cloud_from(x, y, {"value": v}, "device"))
    assert p.last_scatter()["path"] == "binned" and p.last_scatter()["bands_with_scatter"] == 1
    p.finalize()
    check_point_bands(OnHost(p), og, x, y, v, ALL6)
    p.finalize()                                                     # again: still the same bands
    check_point_bands(OnHost(p), og, x, y, v, ALL6)


def test_a_second_ingest_drops_the_stored_bands():
    G, n = 1024, 40_000
    x, y, v = sparse_cloud(G, n, 9)
    og = O.make_grid((0, 0, G, G), tile=(256, 256))
    p = pcr.Pipeline.create(config_for(og, [spec(t) for t in ALL6], scatter_path=2))
    h = n // 2
    p.ingest(cloud_from(x[:h], y[:h], {"value": v[:h]}, "device"))
    assert p.last_scatter()["bands_with_scatter"] == 1
    p.finalize()
    check_point_bands(p, og, x[:h], y[:h], v[:h], ALL6)
    # the second cloud lands in other reference tiles as well: planes AND touched flags change
    x2, y2 = x[h:] + 400.0, y[h:] - 300.0
    p.ingest(cloud_from(x2, y2, {"value": v[h:]}, "device"))
    assert p.last_scatter()["bands_with_scatter"] == 0
    p.finalize()
    check_point_bands(p, og, np.concatenate([x[:h], x2]), np.concatenate([y[:h], y2]), v, ALL6)
    # PipelineConfig.finalize_with_first_ingest = False: never offered
    q = pcr.Pipeline.create(config_for(og, [spec(t) for t in ALL6], scatter_path=2, finalize_with_first_ingest=False))
    q.ingest(cloud_from(x, y, {"value": v}, "device"))
    assert q.last_scatter()["bands_with_scatter"] == 0
    q.finalize()
    check_point_bands(q, og, x, y, v, ALL6)


def test_planes_written_from_outside_drop_the_stored_bands():
    import torch
    G, n = 512, 30_000
    rng = np.random.default_rng(2)
    x, y = rng.uniform(2, G - 2, n), rng.uniform(2, G - 2, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    og = O.make_grid((0, 0, G, G))
    p = pcr.Pipeline.create(config_for(og, [spec("Sum")], scatter_path=2))
    p.ingest(cloud_from(x, y, {"value": v}, "device"))
    assert p.last_scatter()["bands_with_scatter"] == 1
    (ptr, kind, _), = p.state_planes()                               # pointers leave the pipeline ...
    assert p.last_scatter()["bands_with_scatter"] == 0
    t = torch.as_tensor(pcr.DeviceArrayView(ptr, (p.state_row_count(), G), "<f4", owner=p), device="cuda")
    t += 1.0                                                         # ... and the planes are written
    torch.cuda.synchronize()
    p.finalize()
    want = O.run(og, O.SUM, x, y, v, wide=True)
    assert np.allclose(bands(p)[0], want + 1.0, rtol=1e-5, atol=1e-5)
    # and a pipeline whose pointers were handed out BEFORE its first ingest never stores bands with a scatter
    q = pcr.Pipeline.create(config_for(og, [spec("Sum")], scatter_path=2))
    q.tile_touched_ptr()
    q.ingest(cloud_from(x, y, {"value": v}, "device"))
    assert q.last_scatter()["bands_with_scatter"] == 0
    q.finalize()
    assert np.allclose(bands(q)[0], want, rtol=1e-5, atol=1e-5)


def test_touched_flags_written_from_outside_drop_the_stored_bands():
    """The shard exchange ORs the other ranks' touched flags into this rank's (pcr.distributed.allreduce_touched, through
    tile_touched_ptr()): a reference tile nobody touched here becomes 0.0 / NaN-per-cell instead of all NaN.  Bands stored by the
    scatter were made from this rank's flags alone and must not survive."""
    import torch
    G, n = 1024, 40_000
    x, y, v = sparse_cloud(G, n, 21)                                  # one corner: most of the sixteen reference tiles untouched
    og = O.make_grid((0, 0, G, G), tile=(256, 256))
    p = pcr.Pipeline.create(config_for(og, [spec("Sum"), spec("Count")], scatter_path=2))
    p.ingest(cloud_from(x, y, {"value": v}, "device"))
    assert p.last_scatter()["bands_with_scatter"] == 1
    ptr, tx, ty = p.tile_touched_ptr()
    assert p.last_scatter()["bands_with_scatter"] == 0
    flags = torch.as_tensor(pcr.DeviceArrayView(ptr, (ty * tx,), "<i4", owner=p), device="cuda")
    flags.fill_(1)                                                   # "some other rank touched every tile"
    torch.cuda.synchronize()
    p.finalize()
    sm, ct = bands(p)
    want_s = np.nan_to_num(O.run(og, O.SUM, x, y, v, wide=True), nan=0.0)      # Q2: Sum of an empty cell of a touched tile = 0.0
    assert not np.isnan(sm).any() and np.allclose(sm, want_s, rtol=1e-5, atol=1e-5)
    want_c = O.run(og, O.COUNT, x, y, v)
    assert np.array_equal(np.nan_to_num(ct), np.nan_to_num(want_c)) and np.isnan(ct).sum() == (np.nan_to_num(want_c) == 0).sum()


def test_a_split_bin_leaves_the_bands_to_the_finalize_pass():
    """More than 2^17 records in one LDS tile: the tile pass merges that bin with atomics and cannot store bands; the
    device word says so and the finalize kernel runs."""
    G, n = 1024, 600_000
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(500, 510, n - 5000), rng.uniform(2, G - 2, 5000)])
    y = np.concatenate([rng.uniform(500, 510, n - 5000), rng.uniform(2, G - 2, 5000)])
    v = rng.uniform(0, 1, n).astype(np.float32)
    og = O.make_grid((0, 0, G, G), tile=(256, 256))
    poison_device_memory(8 * G * G * 4)
    p = pcr.Pipeline.create(config_for(og, [spec(t) for t in ALL6], scatter_path=2))
    p.ingest(cloud_from(x, y, {"value": v}, "device"))
    assert p.last_scatter()["bands_with_scatter"] == 1               # offered and launched; the device decides
    p.finalize()
    check_point_bands(p, og, x, y, v, ALL6)


def test_two_groups_each_store_their_own_bands():
    G, n = 768, 50_000
    rng = np.random.default_rng(4)
    x, y = rng.uniform(2, 400, n), rng.uniform(2, G - 2, n)
    a = rng.uniform(0, 1, n).astype(np.float32)
    b = rng.uniform(-5, 5, n).astype(np.float32)
    og = O.make_grid((0, 0, G, G), tile=(256, 256))
    specs = [spec("Average", "a"), spec("Max", "b"), spec("Count", "a"), spec("Min", "b")]
    p = pcr.Pipeline.create(config_for(og, specs, scatter_path=2))
    p.ingest(cloud_from(x, y, {"a": a, "b": b}, "device"))
    assert p.last_scatter()["bands_with_scatter"] == 2
    p.finalize()
    got = bands(p)
    for g, (rt, ch, wide) in zip(got, [(O.AVERAGE, a, True), (O.MAX, b, False), (O.COUNT, a, False), (O.MIN, b, False)]):
        want = O.run(og, rt, x, y, ch, wide=wide)
        assert np.array_equal(np.isnan(g), np.isnan(want))
        m = ~np.isnan(want)
        if wide:
            assert (np.abs(g[m].astype(np.float64) - want[m]) <= 1e-5 * np.maximum(1.0, np.abs(want[m]))).all()
        else:
            assert np.array_equal(g[m], want[m])


def test_cabi_word_and_refusals():
    """Through the C-ABI: the device word is 1 after a scatter that stored the bands, finalize_group_unless leaves them alone
    then and overwrites them when the word is 0; a row-block window with a halo does not take the fused form."""
    A = load_cabi()
    L = A.lib()
    W, H, n = 512, 384, 50_000
    rng = np.random.default_rng(1)
    x, y = rng.uniform(0, 200, n), rng.uniform(0, H, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    og = O.make_grid((0.0, 0.0, float(W), float(H)), tile=(128, 128))
    for own, halo, expect in (((0, H), 0, 1), ((128, 256), 8, 0)):
        grid = A.make_grid((0.0, 0.0, float(W), float(H)), dims=(W, H), tile=(128, 128), own_rows=own, halo=halo)
        run = A.ReductionRun(grid, 3, path=2)
        rows = own[1] - own[0]
        try:
            outs = [A.DeviceBuffer(rows * W * 4) for _ in range(3)]
            for o in outs:
                A.check(L.pcr_hip_memset(o.ptr, 0x7F, rows * W * 4, None))
            done = A.DeviceBuffer(4)
            A.check(L.pcr_hip_memset(done.ptr, 0xFF, 4, None))
            rt = (C.c_int * 3)(A.SUM, A.COUNT, A.AVERAGE)
            po = (C.c_void_p * 3)(*[o.ptr.value for o in outs])
            A.check(L.pcr_hip_engine_planes_fresh(run.engine, 2))
            A.check(L.pcr_hip_engine_finalize_with_scatter(run.engine, 3, rt, po, done.ptr))
            run.scatter(x, y, v)
            assert L.pcr_hip_engine_finalize_taken(run.engine) == expect
            word = done.to_numpy(np.uint32, (1,))[0]
            got = [o.to_numpy(np.float32, (rows, W)) for o in outs]
            if expect:
                assert word == 1
                for g, r in zip(got, (O.SUM, O.COUNT, O.AVERAGE)):
                    want = O.run(og, r, x, y, v, wide=r != O.COUNT)
                    assert np.array_equal(np.isnan(g), np.isnan(want))
                    assert np.allclose(g[~np.isnan(want)], want[~np.isnan(want)], rtol=1e-5, atol=1e-6)
                # the hint covered ONE scatter
                run.scatter(x, y, v)
                assert L.pcr_hip_engine_finalize_taken(run.engine) == 0
            else:
                assert word == 0xFFFFFFFF and all((g.view(np.uint32) == 0x7F7F7F7F).all() for g in got)
            # finalize_group_unless: honours the word
            _, tp = run.touched()
            A.check(L.pcr_hip_memset(outs[0].ptr, 0x7F, rows * W * 4, None))
            A.check(L.pcr_hip_memset(done.ptr, 0, 4, None))
            one = (C.c_int * 1)(A.COUNT)
            p1 = (C.c_void_p * 1)(outs[0].ptr.value)
            A.check(L.pcr_hip_finalize_group_unless(C.byref(grid), C.byref(run.planes), tp, 1, one, p1, done.ptr, None))
            A.check(L.pcr_hip_stream_synchronize(None))
            assert not (outs[0].to_numpy(np.float32, (rows, W)).view(np.uint32) == 0x7F7F7F7F).all()
            A.check(L.pcr_hip_memset(outs[0].ptr, 0x7F, rows * W * 4, None))
            A.check(L.pcr_hip_memset(done.ptr, 1, 4, None))
            A.check(L.pcr_hip_finalize_group_unless(C.byref(grid), C.byref(run.planes), tp, 1, one, p1, done.ptr, None))
            A.check(L.pcr_hip_stream_synchronize(None))
            assert (outs[0].to_numpy(np.float32, (rows, W)).view(np.uint32) == 0x7F7F7F7F).all()
        finally:
            run.close()


def test_finalize_async_leaves_stream_ordered_device_bands():
    """Pipeline.finalize_async (extension): with a device-resident result the finalize kernels are only enqueued; the bands
    are complete after synchronize().  Several pipelines back to back, as bench.py's timed steps run them; a host-resident
    result makes the call an ordinary finalize."""
    G, n = 1024, 300_000
    rng = np.random.default_rng(13)
    x, y = rng.uniform(2, G - 2, n), rng.uniform(2, 700, n)
    v = rng.uniform(-1, 1, n).astype(np.float32)
    og = O.make_grid((0, 0, G, G), tile=(256, 256))
    cloud = cloud_from(x, y, {"value": v}, "device")
    pipes = []
    for _ in range(4):
        cfg = config_for(og, [spec(t) for t in ALL6], scatter_path=2)
        cfg.result_location = pcr.MemoryLocation.Device
        p = pcr.Pipeline.create(cfg)
        p.ingest(cloud)
        p.ingest(cloud)                                            # (the second ingest: the ordinary finalize kernel, enqueued)
        p.finalize_async()
        assert p.result() is not None
        pipes.append(p)
    for p in pipes:
        p.synchronize()
        check_point_bands(OnHost(p), og, np.concatenate([x, x]), np.concatenate([y, y]), np.concatenate([v, v]), ALL6)
    q = pcr.Pipeline.create(config_for(og, [spec("Count")], scatter_path=2))      # host-resident result
    q.ingest(cloud)
    q.finalize_async()
    want = O.run(og, O.COUNT, x, y, v)
    assert np.array_equal(bands(q)[0], want, equal_nan=True)


def test_merge_touched_keeps_the_stored_bands_unless_a_flag_changes():
    """Row-block shards without glyph planes all-reduce their touched flags in a copy and hand the union to
    Pipeline.merge_touched: an unchanged union (the usual case) leaves the bands the scatter stored valid; a tile another rank
    touched makes the device drop them and the finalize pass runs with the merged flags."""
    import torch
    G, n = 1024, 40_000
    x, y, v = sparse_cloud(G, n, 33)
    og = O.make_grid((0, 0, G, G), tile=(256, 256))
    want_s = O.run(og, O.SUM, x, y, v, wide=True)
    for other_rank_touches_more in (False, True):
        p = pcr.Pipeline.create(config_for(og, [spec("Sum"), spec("Count")], scatter_path=2))
        p.ingest(cloud_from(x, y, {"value": v}, "device"))
        ptr, tx, ty = p.tile_touched_ptr(readonly=True)
        assert p.last_scatter()["bands_with_scatter"] == 1               # reading the flags drops nothing
        mine = torch.as_tensor(pcr.DeviceArrayView(ptr, (ty * tx,), "<i4", owner=p), device="cuda")
        union = mine.clone()
        if other_rank_touches_more:
            union.fill_(1)
        torch.cuda.synchronize()
        p.merge_touched(union.data_ptr())
        assert p.last_scatter()["bands_with_scatter"] == 1               # (the device decides)
        p.finalize()
        sm, ct = bands(p)
        if other_rank_touches_more:
            assert not np.isnan(sm).any() and np.allclose(sm, np.nan_to_num(want_s, nan=0.0), rtol=1e-5, atol=1e-5)
            assert int(mine.sum().item()) == tx * ty
        else:
            assert np.array_equal(np.isnan(sm), np.isnan(want_s)) and np.isnan(sm).any()
            m = ~np.isnan(want_s)
            assert np.allclose(sm[m], want_s[m], rtol=1e-5, atol=1e-5)
        assert np.array_equal(np.nan_to_num(ct), np.nan_to_num(O.run(og, O.COUNT, x, y, v)))


def test_two_level_sweep_defines_the_planes_and_stores_the_bands(monkeypatch):
    """A window with more LDS tiles than one binning pass takes (PCR_HIP_DEBUG_MAX_BINS lowers the limit) is sorted in two
    levels; its tile pass, too, defines every cell of undefined planes and stores the bands -- poisoned memory, a sparse
    cloud (empty tiles get their identity from empty work items), sixteen reference tiles."""
    monkeypatch.setenv("PCR_HIP_DEBUG_MAX_BINS", "12")          # read by pcr_hip_engine_create
    monkeypatch.setenv("PCR_HIP_DEBUG_TWO_LEVEL", "1")
    G, n = 1024, 60_000
    x, y, v = sparse_cloud(G, n, 17)
    og = O.make_grid((0, 0, G, G), tile=(256, 256))
    poison_device_memory(8 * G * G * 4)
    p = pcr.Pipeline.create(config_for(og, [spec(t) for t in ALL6], scatter_path=2))
    p.ingest(cloud_from(x, y, {"value": v}, "device"))
    info = p.last_scatter()
    assert info["path"] == "binned" and info["num_bins"] > 12 and info["bands_with_scatter"] == 1
    p.finalize()
    check_point_bands(p, og, x, y, v, ALL6)
    # and a second ingest on top of what the two-level sweep defined
    p.ingest(cloud_from(x + 300.0, y - 200.0, {"value": v}, "device"))
    p.finalize()
    check_point_bands(p, og, np.concatenate([x, x + 300.0]), np.concatenate([y, y - 200.0]), np.concatenate([v, v]), ALL6)
