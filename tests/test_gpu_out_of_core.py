"""Out-of-core grids (second half): a grid whose accumulation state exceeds `gpu_memory_budget` is swept in
row bands of whole reference-tile rows -- one band's planes in HBM at a time, the others parked in host memory up to
`host_cache_budget` and, beyond it, as the reference's own `.pcrt` tile files under `state_dir` (round 5: the reference's
spill IS its checkpoint format, src/engine/tile_manager.cpp:76-138 -> src/io/tile_state_io.cpp:45-95; an out-of-core pipeline
can therefore save_state / load_state / resume like any other).  Replaces the reference's TileManager LRU + disk spill
(src/engine/tile_manager.cpp:76-138 evict + flush, :183-375 acquire with disk load).  Footprints are clipped to the
reference tile of their centre cell (Q4), so a band is a shard without a halo and the results must equal the in-core
pipeline's -- bit for bit for everything that does not go through float atomics (Point planes, Line counts), to fp32
re-association for the Gaussian sums -- and the oracle's."""
import ctypes as C
import os

import numpy as np
import pytest

import pcr
import pcr_oracle_py as O
from conftest import assert_band_close
from test_gpu_pipeline_api import cloud_from, config_for, spec

pytestmark = pytest.mark.gpu


def specs():
    out = [spec(t) for t in ("Sum", "Count", "Average", "Max", "Min")]
    out.append(pcr.gaussian_splat_spec("value", default_sigma=1.0, max_radius_cells=4.0))
    ln = pcr.line_splat_spec("value", default_direction=0.4, default_half_length=6.0, max_radius_cells=8.0)
    ln.type = pcr.ReductionType.Count
    out.append(ln)
    return out


def bands_of(p):
    return [np.array(p.result().band_array(i)) for i in range(p.result().num_bands())]


def clouds(G, seed):
    rng = np.random.default_rng(seed)
    out = []
    for n in (150_000, 90_000):
        x, y = rng.uniform(0, G, n), rng.uniform(0, G, n)
        x[:500], y[:500] = rng.uniform(250, 262, 500), rng.uniform(G - 262, G - 250, 500)      # points on both sides of a tile corner
        out.append((x, y, rng.uniform(-1, 1, n).astype(np.float32)))
    return out


@pytest.mark.parametrize("host_cache", [1, 1 << 30], ids=["spilled_to_disk", "parked_in_host_memory"])
def test_out_of_core_equals_in_core_bit_for_bit(tmp_path, host_cache):
    G = 1024
    og = O.make_grid((0, 0, G, G), tile=(256, 256))
    cl = clouds(G, 3)
    incore = pcr.Pipeline.create(config_for(og, specs()))
    assert incore is not None and not incore.out_of_core()
    # 8 planes + 7 bands per cell = 60 KB per row: one 256-row tile row is 15 MB -- a 3 MB budget forces one tile row per band
    ooc = pcr.Pipeline.create(config_for(og, specs(), gpu_memory_budget=3 << 20, host_cache_budget=host_cache, state_dir=str(tmp_path)))
    assert ooc is not None, pcr.pipeline_create_error()
    assert ooc.out_of_core()
    for x, y, v in cl:
        c = cloud_from(x, y, {"value": v}, "host")
        incore.ingest(c)
        ooc.ingest(c)
    spill = ooc.spill_dir()
    assert os.path.dirname(spill) == str(tmp_path) and os.listdir(tmp_path) == [os.path.basename(spill)]
    if host_cache == 1:
        # every band was evicted: seven reductions -> reduction_0 .. reduction_6, each with the 16 touched tiles of the grid
        assert sorted(os.listdir(spill)) == [f"reduction_{r}" for r in range(7)]
        names = sorted(os.listdir(os.path.join(spill, "reduction_2")))
        assert names == [f"tile_{r:04d}_{c:04d}.pcrt" for r in range(4) for c in range(4)]
        row, col, state, rtype = pcr.read_tile_state(os.path.join(spill, "reduction_2", "tile_0001_0003.pcrt"))
        assert (row, col, rtype) == (1, 3, pcr.ReductionType.Average) and state.shape == (2, 256, 256)
    else:
        assert os.listdir(spill) == []
    incore.finalize()
    ooc.finalize()
    a, b = bands_of(incore), bands_of(ooc)
    assert len(a) == len(b) == 7
    for k, (u, w) in enumerate(zip(a, b)):
        assert u.shape == w.shape == (G, G)
        if k == 5:
            # the Gaussian tiles merge their windows into the planes with float atomics: the order of those adds is not
            # reproducible between two runs of the SAME pipeline either -- equal up to fp32 re-association, same NaN mask
            assert np.array_equal(np.isnan(u), np.isnan(w))
            m = ~np.isnan(u)
            assert (np.abs(u[m] - w[m]) <= 1e-6 + 1e-5 * np.abs(u[m])).all(), "Gaussian band: out of core vs in core"
        else:
            assert np.array_equal(u, w, equal_nan=True), f"band {k}: out of core != in core"
    # and against the oracle (Count bit-exact)
    x = np.concatenate([c[0] for c in cl]); y = np.concatenate([c[1] for c in cl]); v = np.concatenate([c[2] for c in cl])
    assert np.array_equal(b[1], O.run(og, O.COUNT, x, y, v), equal_nan=True)
    want_line = O.run(og, O.COUNT, x, y, v, glyph=O.make_glyph(O.GLYPH_LINE, direction=0.4, half_length=6.0, max_radius=8.0))
    assert np.array_equal(b[6], want_line, equal_nan=True)
    st = ooc.stats()
    assert st.points_processed == 240_000 and st.tiles_active > 0


def test_untouched_bands_stay_nan_and_cost_no_parking(tmp_path):
    """Points in the top tile row only: the other bands are never parked (nothing fell there) and finalize to NaN (Q3)."""
    G = 1024
    og = O.make_grid((0, 0, G, G), tile=(256, 256))
    rng = np.random.default_rng(9)
    n = 50_000
    x, y = rng.uniform(0, G, n), rng.uniform(G - 200, G, n)          # rows 0..199
    v = rng.uniform(0, 1, n).astype(np.float32)
    p = pcr.Pipeline.create(config_for(og, [spec("Sum")], gpu_memory_budget=1 << 20, host_cache_budget=1, state_dir=str(tmp_path)))
    assert p is not None and p.out_of_core()
    p.ingest(cloud_from(x, y, {"value": v}, "device"))
    assert p.last_scatter()["points_valid"] == n                       # over all bands (three of the four were not even visited)
    assert sorted(os.listdir(p.spill_dir())) == [f"tile_0000_{c:04d}.pcrt" for c in range(4)]  # one reduction: the reference's flat layout
    p.finalize()
    got = bands_of(p)[0]
    assert np.isnan(got[256:]).all()
    assert_band_close(got, O.run(og, O.SUM, x, y, v, wide=True), rtol=1e-5, atol=1e-6, what="out-of-core Sum")


def test_out_of_core_needs_a_host_result(tmp_path):
    G = 512
    og = O.make_grid((0, 0, G, G), tile=(128, 128))
    cfg = config_for(og, [spec("Sum")], gpu_memory_budget=1 << 18, result_location=pcr.MemoryLocation.Device)
    assert pcr.Pipeline.create(cfg) is None
    assert "out of core" in pcr.pipeline_create_error() and "result_location = Host" in pcr.pipeline_create_error()
    p = pcr.Pipeline.create(config_for(og, [spec("Sum")], gpu_memory_budget=1 << 18))
    assert p is not None and p.out_of_core() and p.engine() == "hip"
    assert p.state_planes() == [] and p.halo_rows() == 0


@pytest.mark.skipif(O.ref_lib() is None, reason="oracle/_ref not built (reference tree absent)")
def test_spilled_bands_are_read_by_the_reference_reader(tmp_path):
    """What an out-of-core pipeline evicts are the reference's tile files: the reference's OWN reader (oracle/_ref, built from
    src/io/tile_state_io.cpp) takes every one of them, and the state inside is the oracle's state of that tile."""
    G = 512
    og = O.make_grid((0, 0, G, G), tile=(128, 128))
    rng = np.random.default_rng(4)
    n = 80_000
    x, y = rng.uniform(0, G, n), rng.uniform(0, G, n)
    v = rng.uniform(0, 4, n).astype(np.float32)
    p = pcr.Pipeline.create(config_for(og, [spec("Average")], gpu_memory_budget=1 << 18, host_cache_budget=1, state_dir=str(tmp_path)))
    assert p is not None and p.out_of_core()
    p.ingest(cloud_from(x, y, {"value": v}, "host"))
    spill = p.spill_dir()
    files = sorted(os.listdir(spill))
    assert files == [f"tile_{r:04d}_{c:04d}.pcrt" for r in range(4) for c in range(4)]
    R = O.ref_lib()
    R.pcr_ref_read_tile_state.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 6 + [C.c_void_p]
    count = np.nan_to_num(O.run(og, O.COUNT, x, y, v))
    total = O.run(og, O.SUM, x, y, v, wide=True)
    for r in range(4):
        for c in range(4):
            hdr = [C.c_int(0) for _ in range(6)]
            st = np.zeros((2, 128, 128), np.float32)
            path = os.path.join(spill, f"tile_{r:04d}_{c:04d}.pcrt").encode()
            assert R.pcr_ref_read_tile_state(path, *[C.byref(a) for a in hdr], st.ctypes.data) == 0
            assert [a.value for a in hdr] == [r, c, 128, 128, 2, int(pcr.ReductionType.Average)]
            rows, cols = slice(128 * r, 128 * (r + 1)), slice(128 * c, 128 * (c + 1))
            assert np.array_equal(st[1], count[rows, cols])                                   # {sum, count}: builtin_ops.h:62-72
            assert (np.abs(st[0] - total[rows, cols]) <= 1e-5 * np.maximum(1.0, np.abs(total[rows, cols]))).all()
    del p
    assert os.listdir(tmp_path) == []            # a spill is working state: the pipeline takes its directory with it


@pytest.mark.parametrize("host_cache", [1, 1 << 30], ids=["spilled_to_disk", "parked_in_host_memory"])
def test_out_of_core_checkpoint_and_resume_equal_the_uninterrupted_run(tmp_path, host_cache):
    """save_state of an out-of-core pipeline, resume in a NEW out-of-core pipeline and in an IN-CORE one, ingest the second
    cloud: all equal one pipeline that saw both clouds (Point bands and Line counts bit for bit)."""
    G = 1024
    og = O.make_grid((0, 0, G, G), tile=(256, 256))
    cl = clouds(G, 17)
    reds = lambda: [spec(t) for t in ("Sum", "Count", "Average", "Max", "Min")]          # noqa: E731
    whole = pcr.Pipeline.create(config_for(og, reds()))
    for x, y, v in cl:
        whole.ingest(cloud_from(x, y, {"value": v}, "host"))
    whole.finalize()
    want = bands_of(whole)
    work, ck = str(tmp_path / "work"), str(tmp_path / "ck")
    a = pcr.Pipeline.create(config_for(og, reds(), gpu_memory_budget=2 << 20, host_cache_budget=host_cache, state_dir=work))
    assert a.out_of_core()
    a.ingest(cloud_from(*cl[0][:2], {"value": cl[0][2]}, "host"))
    a.save_state(ck)                                         # a separate directory: the complete state, every band
    assert sorted(os.listdir(ck)) == [f"reduction_{r}" for r in range(5)]
    assert len(os.listdir(os.path.join(ck, "reduction_0"))) == 16
    a.save_state()                                           # ... and into its own state_dir: a checkpoint that outlives it
    del a
    assert sorted(os.listdir(work)) == [f"reduction_{r}" for r in range(5)] and len(os.listdir(os.path.join(work, "reduction_3"))) == 16
    x, y, v = cl[1]
    for kw in (dict(gpu_memory_budget=2 << 20, host_cache_budget=host_cache, state_dir=ck, resume=True),        # out of core again
               dict(state_dir=work, resume=True)):                                                               # in core
        b = pcr.Pipeline.create(config_for(og, reds(), **kw))
        assert b is not None, pcr.pipeline_create_error()
        assert b.out_of_core() == ("gpu_memory_budget" in kw)
        b.ingest(cloud_from(x, y, {"value": v}, "device"))
        b.finalize()
        for k, (u, w) in enumerate(zip(want, bands_of(b))):
            assert np.array_equal(u, w, equal_nan=True), (k, kw)
