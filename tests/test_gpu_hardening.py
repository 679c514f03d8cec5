"""Round-2 hardening of the engine around the hot path:
  * a row-block shard never clips a Line silently: per-point half_length beyond the halo is refused, with the halo
    raised (PipelineConfig.shard_halo_rows) two shards equal the unsharded oracle (glyph_kernels.cu:228-234, Q7);
  * the device-wide scratch arena is borrowed exclusively: two host threads driving two pipelines on one device;
  * the C-ABI arena (MemoryPool contract, memory_pool.cu:24-59) and the engine scratch that is made of it;
  * the routing key of the last tile cannot collide with the dropped-point sentinel."""
import ctypes as C
import threading

import numpy as np
import pytest

import pcr
import pcr_oracle_py as O
from conftest import assert_band_close, load_cabi
from test_gpu_pipeline_api import cloud_from, config_for, spec

pytestmark = pytest.mark.gpu


def _line_inputs(G, n, hl_max):
    rng = np.random.default_rng(31)
    x, y = rng.uniform(0, G, n), rng.uniform(0, G, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    d = rng.uniform(0, np.pi, n).astype(np.float32)
    hl = rng.uniform(1.0, hl_max, n).astype(np.float32)
    return x, y, v, d, hl


def test_sharded_line_with_per_point_half_length_is_refused_or_exact():
    A = load_cabi()
    L = A.lib()
    G = 128
    og = O.make_grid((0, 0, G, G))                       # one 4096-tile: the blocks below cut it
    x, y, v, d, hl = _line_inputs(G, 6000, 20.0)          # reach up to 21 rows, max_radius 4 caps x only (csy < 0)
    ls = pcr.line_splat_spec("value", direction_channel="direction", half_length_channel="half_length",
                             default_half_length=2.0, max_radius_cells=4.0)
    ls.type = pcr.ReductionType.Sum
    cloud = cloud_from(x, y, {"value": v, "direction": d, "half_length": hl}, "device")
    blocks = ((0, 50), (50, 128))
    # default halo = max(|default_half_length / csy|, max_radius) + 1 = 5 rows: too small for this cloud -> loud error
    p = pcr.Pipeline.create(config_for(og, [ls], shard_row_begin=0, shard_row_end=50))
    assert p.halo_rows() == 5
    with pytest.raises(RuntimeError, match="shard_halo_rows >= 21"):
        p.ingest(cloud)
    # whole-grid pipelines and tile-aligned blocks never need the check
    whole = pcr.Pipeline.create(config_for(og, [ls]))
    whole.ingest(cloud)
    whole.finalize()
    want = O.run(og, O.SUM, x, y, v, glyph=O.make_glyph(O.GLYPH_LINE, half_length=2.0, max_radius=4.0),
                 direction=d, half_length=hl)
    assert_band_close(np.array(whole.result().band_array(0)), want, rtol=1e-5, atol=1e-5, what="unsharded line")
    # with the halo raised: two shards + halo merge == the unsharded oracle
    shards = []
    for r0, r1 in blocks:
        q = pcr.Pipeline.create(config_for(og, [ls], shard_row_begin=r0, shard_row_end=r1, shard_halo_rows=21))
        assert q.halo_rows() == 21
        q.ingest(cloud)
        q.synchronize()
        shards.append(q)
    top, bot = shards
    halo, row = 21, G * 4
    assert bot.state_row_begin() == 50 - halo and top.state_row_count() == 50 + halo
    for (tp, kind, _), (bp, _, _) in zip(top.state_planes(), bot.state_planes()):
        # top's apron rows [50, 71) -> bot's own rows; bot's apron rows [29, 50) -> top's own rows
        A.check(L.pcr_hip_plane_merge(kind, C.c_void_p(bp + halo * row), C.c_void_p(tp + 50 * row), halo * G, None))
        A.check(L.pcr_hip_plane_merge(kind, C.c_void_p(tp + (50 - halo) * row), C.c_void_p(bp), halo * G, None))
    A.check(L.pcr_hip_device_synchronize())
    for q in shards:
        q.finalize()
    got = np.vstack([np.array(top.result().band_array(0)), np.array(bot.result().band_array(0))])
    assert_band_close(got, want, rtol=1e-5, atol=1e-5, what="two shards, per-point half_length, raised halo")


def test_sharded_line_reach_keeps_the_sign_of_half_length():
    """The reference caps hy = half_length / cell_size_y with std::min(hy, cap) (glyph_kernels.cu:228-234): only the sign of
    half_length that makes hy POSITIVE is capped by max_radius_cells; the other one reaches |half_length / cell_size_y| rows.
    The shard's reach check must tell the two apart (it reduced max |hl| and lost the sign)."""
    G, n = 128, 4000
    rng = np.random.default_rng(5)
    x, y = rng.uniform(30, G - 30, n), rng.uniform(30, G - 30, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    d = rng.uniform(0, np.pi, n).astype(np.float32)
    hl_pos = rng.uniform(10.0, 20.0, n).astype(np.float32)
    hl_neg = (-hl_pos).astype(np.float32)
    ls = pcr.line_splat_spec("value", direction_channel="direction", half_length_channel="half_length",
                             default_half_length=2.0, max_radius_cells=4.0)
    ls.type = pcr.ReductionType.Sum
    far = int(np.ceil(float(hl_pos.max()))) + 1

    def reach(og, hl):
        p = pcr.Pipeline.create(config_for(og, [ls], shard_row_begin=0, shard_row_end=50))
        assert p.halo_rows() == 5
        return p.line_reach_rows(cloud_from(x, y, {"value": v, "direction": d, "half_length": hl}, "device"))

    north, south = O.make_grid((0, 0, G, G)), O.make_grid((0, 0, G, G), cell=(1.0, 1.0))
    assert reach(north, hl_pos) == far          # hy < 0: never capped
    assert reach(north, hl_neg) == 5            # hy > 0: capped at max_radius_cells = 4 (+ 1 for the end points' rounding)
    assert reach(south, hl_pos) == 5
    assert reach(south, hl_neg) == far
    # the capped side really is capped: two north-up shards with the DEFAULT halo of 5 rows add up to the unsharded oracle
    A = load_cabi()
    L = A.lib()
    cloud = cloud_from(x, y, {"value": v, "direction": d, "half_length": hl_neg}, "device")
    want = O.run(north, O.SUM, x, y, v, glyph=O.make_glyph(O.GLYPH_LINE, half_length=2.0, max_radius=4.0),
                 direction=d, half_length=hl_neg)
    shards = []
    for r0, r1 in ((0, 50), (50, 128)):
        q = pcr.Pipeline.create(config_for(north, [ls], shard_row_begin=r0, shard_row_end=r1))
        q.ingest(cloud)
        q.synchronize()
        shards.append(q)
    top, bot = shards
    halo, row = 5, G * 4
    for (tp, kind, _), (bp, _, _) in zip(top.state_planes(), bot.state_planes()):
        A.check(L.pcr_hip_plane_merge(kind, C.c_void_p(bp + halo * row), C.c_void_p(tp + 50 * row), halo * G, None))
        A.check(L.pcr_hip_plane_merge(kind, C.c_void_p(tp + (50 - halo) * row), C.c_void_p(bp), halo * G, None))
    A.check(L.pcr_hip_device_synchronize())
    for q in shards:
        q.finalize()
    got = np.vstack([np.array(top.result().band_array(0)), np.array(bot.result().band_array(0))])
    assert_band_close(got, want, rtol=1e-5, atol=1e-5, what="two shards, negative half_length (capped side), default halo")


def test_two_threads_two_pipelines_share_the_scratch_arena_safely():
    """the arena was only safe for one host thread.  Two threads hammer two pipelines on one device
    (binned path: both borrow the device-wide scratch on every ingest); every result must equal the oracle."""
    G, n, rounds = 512, 300_000, 6
    og = O.make_grid((0, 0, G, G))
    rng = np.random.default_rng(3)
    data = []
    for t in range(2):
        x, y = rng.uniform(0, G, n), rng.uniform(0, G, n)
        v = rng.uniform(0, 1, n).astype(np.float32)
        data.append((x, y, v, cloud_from(x, y, {"value": v}, "device")))
    results, errors = [None, None], []

    def work(t):
        try:
            x, y, v, cloud = data[t]
            outs = []
            for _ in range(rounds):
                p = pcr.Pipeline.create(config_for(og, [spec("Count"), spec("Max"), spec("Sum")], scatter_path=2,
                                                   result_location=pcr.MemoryLocation.Host))
                p.ingest(cloud)
                p.finalize()
                outs.append([np.array(p.result().band_array(b)) for b in range(3)])
            results[t] = outs
        except Exception as exc:                 # surfaced in the main thread
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(2):
        x, y, v, _ = data[t]
        wc, wm, ws = O.run(og, O.COUNT, x, y, v), O.run(og, O.MAX, x, y, v), O.run(og, O.SUM, x, y, v, wide=True)
        for outs in results[t]:
            assert_band_close(outs[0], wc, what=f"thread {t} count")
            assert_band_close(outs[1], wm, what=f"thread {t} max")
            assert_band_close(outs[2], ws, rtol=1e-5, atol=1e-5, what=f"thread {t} sum")


def test_arena_contract_and_engine_scratch_is_an_arena():
    A = load_cabi()
    L = A.lib()
    a = C.c_void_p()
    A.check(L.pcr_hip_arena_create(C.byref(a), 1 << 20))
    p1, p2, p3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    A.check(L.pcr_hip_arena_alloc(a, 1000, C.byref(p1)))
    A.check(L.pcr_hip_arena_alloc(a, 10, C.byref(p2)))
    assert p1.value % 256 == 0 and p2.value % 256 == 0 and p2.value == p1.value + 1024       # 256-B aligned bump
    assert L.pcr_hip_arena_alloc(a, 1 << 20, C.byref(p3)) == 2 and p3.value is None            # OutOfMemory, loud
    assert b"pool exhausted" in L.pcr_hip_last_error()
    cap, used, hw = C.c_size_t(), C.c_size_t(), C.c_size_t()
    A.check(L.pcr_hip_arena_stats(a, C.byref(cap), C.byref(used), C.byref(hw)))
    assert cap.value == 1 << 20 and used.value == 1034 and hw.value == 1034
    # the memory is real device memory
    A.check(L.pcr_hip_memset(p1, 0x5A, 1000, None))
    host = np.zeros(1000, dtype=np.uint8)
    A.check(L.pcr_hip_memcpy_d2h(host.ctypes.data, p1, 1000, None))
    A.check(L.pcr_hip_stream_synchronize(None))
    assert (host == 0x5A).all()
    A.check(L.pcr_hip_arena_reset(a))
    A.check(L.pcr_hip_arena_alloc(a, 16, C.byref(p3)))
    assert p3.value == p1.value
    A.check(L.pcr_hip_arena_stats(a, C.byref(cap), C.byref(used), C.byref(hw)))
    assert used.value == 16 and hw.value == 1034
    A.check(L.pcr_hip_arena_destroy(a))

    # the engine's scratch: pre-sized at create, borrowed once per binned scatter, never re-allocated in steady state
    G, n = 1024, 400_000
    og = O.make_grid((0, 0, G, G))
    rng = np.random.default_rng(8)
    x, y = rng.uniform(0, G, n), rng.uniform(0, G, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    cloud = cloud_from(x, y, {"value": v}, "device")

    def stats():
        c, h, b, g = C.c_size_t(), C.c_size_t(), C.c_uint64(), C.c_uint64()
        A.check(L.pcr_hip_device_scratch_stats(0, C.byref(c), C.byref(h), C.byref(b), C.byref(g)))
        return c.value, h.value, b.value, g.value

    p = pcr.Pipeline.create(config_for(og, [spec("Sum")], scatter_path=2, gpu_pool_size_bytes=64 << 20))
    cap0, _, b0, g0 = stats()
    assert cap0 >= 64 << 20
    for _ in range(3):
        p.ingest(cloud)
    p.finalize()
    cap1, hw1, b1, g1 = stats()
    assert b1 == b0 + 3 and g1 == g0 and cap1 == cap0                  # three borrows, no growth
    assert 12 * n <= hw1 <= cap1                                       # keys 4 B + records 8 B per point were carved from it
    assert_band_close(np.array(p.result().band_array(0)), 3 * O.run(og, O.SUM, x, y, v, wide=True), rtol=1e-5, atol=1e-5,
                      what="three ingests")


def test_last_tile_key_cannot_collide_with_the_sentinel():
    """Routing key = tile << 15 | local cell, 0xFFFFFFFF = dropped point: a grid with exactly 2^17 LDS tiles must not
    take the two-level sort.  Reached on a small grid by shrinking what one pass may count."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import pcr
# Count only: LDS tiles of 128 x 128 cells; W = 128 * 512, H = 128 * 256 -> exactly 131072 = 2^17 tiles
W, H = 128 * 512, 128 * 256
cfg = pcr.PipelineConfig()
cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(W), float(H)); cfg.grid.compute_dimensions()
cfg.exec_mode = pcr.ExecutionMode.GPU
r = pcr.ReductionSpec(); r.value_channel, r.type = "value", pcr.ReductionType.Count
cfg.reductions = [r]; cfg.result_location = pcr.MemoryLocation.Device
cfg.scatter_path = 2                                   # binned: the sparse cloud would otherwise take the direct path
n = 2_000_000
rng = np.random.default_rng(1)
x = rng.uniform(0, W, n); y = rng.uniform(0, H, n)
x[:1000] = W - 0.5; y[:1000] = 0.25                    # the last cell of the last tile
v = np.ones(n, dtype=np.float32)
c = pcr.PointCloud.create(n); c.set_x_array(x); c.set_y_array(y); c.add_channel("value", pcr.DataType.Float32); c.set_channel_array_f32("value", v)
p = pcr.Pipeline.create(cfg); assert p is not None, pcr.pipeline_create_error()
p.ingest(c.to_device()); p.finalize()
res = p.result()
band = torch.as_tensor(pcr.DeviceArrayView(res.band_device_ptr(0), (H, W), "<f4", owner=res), device="cuda")
total = float(torch.nansum(band, dtype=torch.float64).item())
assert total == n, total
assert float(band[H - 1, W - 1].item()) >= 1000
print("OK", p.last_scatter())
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code, os.path.join(root, "pointcloud-raster_amd", "python")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-1000:] + out.stderr[-3000:]
