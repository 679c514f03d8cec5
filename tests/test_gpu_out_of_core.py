"""Out-of-core grids (second half): a grid whose accumulation state exceeds `gpu_memory_budget` is swept in
row bands of whole reference-tile rows -- one band's planes in HBM at a time, the others parked in host memory up to
`host_cache_budget` and in files under `state_dir` beyond it.  Replaces the reference's TileManager LRU + disk spill
(src/engine/tile_manager.cpp:76-138 evict + flush, :183-375 acquire with disk load).  Footprints are clipped to the
reference tile of their centre cell (Q4), so a band is a shard without a halo and the results must equal the in-core
pipeline's -- bit for bit for everything that does not go through float atomics (Point planes, Line counts), to fp32
re-association for the Gaussian sums -- and the oracle's."""
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
    if host_cache == 1:
        assert sorted(os.listdir(tmp_path)) == [f"band_{b}.state" for b in range(4)]            # every band was evicted
    else:
        assert os.listdir(tmp_path) == []
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
    assert os.listdir(tmp_path) == ["band_0.state"]
    p.finalize()
    got = bands_of(p)[0]
    assert np.isnan(got[256:]).all()
    assert_band_close(got, O.run(og, O.SUM, x, y, v, wide=True), rtol=1e-5, atol=1e-6, what="out-of-core Sum")


def test_out_of_core_needs_a_host_result_and_refuses_checkpoints(tmp_path):
    G = 512
    og = O.make_grid((0, 0, G, G), tile=(128, 128))
    cfg = config_for(og, [spec("Sum")], gpu_memory_budget=1 << 18, result_location=pcr.MemoryLocation.Device)
    assert pcr.Pipeline.create(cfg) is None
    assert "out of core" in pcr.pipeline_create_error() and "result_location = Host" in pcr.pipeline_create_error()
    p = pcr.Pipeline.create(config_for(og, [spec("Sum")], gpu_memory_budget=1 << 18))
    assert p is not None and p.out_of_core()
    with pytest.raises(RuntimeError, match="out-of-core"):
        p.save_state(str(tmp_path))
    assert p.state_planes() == [] and p.halo_rows() == 0
