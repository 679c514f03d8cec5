"""BASELINE.json's full sizes (C2: 50 M points, 4096 x 4096) checked through size-independent
properties -- the oracle would need minutes here: conservation of counts and sums, Average =
Sum / Count cell by cell, min <= avg <= max, permutation invariance of Count/Min/Max (bit-exact)
and of Sum (tolerance), linearity in the values, idempotent re-finalize, additivity over split
ingests, path agreement (direct atomics vs binned LDS tiles)."""
import numpy as np
import pytest

import pcr

pytestmark = pytest.mark.gpu
G, N = 4096, 50_000_000


def _spec(t, ch="value"):
    r = pcr.ReductionSpec()
    r.value_channel, r.type = ch, t
    return r


def _cfg(reductions, path=0):
    cfg = pcr.PipelineConfig()
    cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(G))
    cfg.grid.compute_dimensions()
    cfg.exec_mode = pcr.ExecutionMode.GPU
    cfg.reductions = reductions
    cfg.scatter_path = path
    return cfg


@pytest.fixture(scope="module")
def data():
    rng = np.random.default_rng(42)
    x = rng.uniform(2, G - 2, N)
    y = rng.uniform(2, G - 2, N)
    v = rng.uniform(0, 1, N).astype(np.float32)
    c = pcr.PointCloud.create(N)
    c.set_x_array(x)
    c.set_y_array(y)
    c.add_channel("value", pcr.DataType.Float32)
    c.set_channel_array_f32("value", v)
    c.add_channel("twice", pcr.DataType.Float32)
    c.set_channel_array_f32("twice", 2.0 * v)
    return dict(x=x, y=y, v=v, dev=c.to_device())


def _bands(pipe):
    return [np.array(pipe.result().band_array(i)) for i in range(pipe.result().num_bands())]


def test_c2_conservation_and_consistency(data):
    T = pcr.ReductionType
    pipe = pcr.Pipeline.create(_cfg([_spec(T.Sum), _spec(T.Count), _spec(T.Average), _spec(T.Min), _spec(T.Max),
                                     _spec(T.Sum, "twice")]))
    pipe.ingest(data["dev"])
    pipe.finalize()
    s, c, a, mn, mx, s2 = _bands(pipe)
    info = pipe.last_scatter()
    assert info["path"] == "binned" and info["points_valid"] == N
    assert np.nansum(c.astype(np.float64)) == N                       # every point counted exactly once
    assert abs(s.astype(np.float64).sum() - data["v"].astype(np.float64).sum()) < 1e-6 * N
    occ = ~np.isnan(c)
    assert np.array_equal(occ, ~np.isnan(a)) and np.array_equal(occ, ~np.isnan(mn))
    assert (s[~occ] == 0).all()                                       # Q2: empty cell of a touched tile sums to 0
    np.testing.assert_allclose(a[occ], s[occ] / c[occ], rtol=2e-7)    # Average is the same two planes
    assert (mn[occ] <= a[occ] * (1 + 1e-6)).all() and (a[occ] <= mx[occ] * (1 + 1e-6)).all()
    np.testing.assert_allclose(s2, 2.0 * s, rtol=1e-6, atol=1e-6)     # linearity in the values (power of 2: near exact)
    # idempotent: finalize again without ingest gives identical bands (state survives finalize)
    pipe.finalize()
    again = _bands(pipe)
    assert np.array_equal(again[0], s) and np.array_equal(again[1], c, equal_nan=True)


def test_c2_paths_and_permutation_agree(data):
    T = pcr.ReductionType
    reds = [_spec(T.Sum), _spec(T.Count), _spec(T.Min), _spec(T.Max)]
    ref = pcr.Pipeline.create(_cfg(reds, path=2))
    ref.ingest(data["dev"])
    ref.finalize()
    s, c, mn, mx = _bands(ref)
    # direct global atomics on a permuted, split cloud
    perm = np.random.default_rng(1).permutation(N)
    half = N // 2
    other = pcr.Pipeline.create(_cfg(reds, path=1))
    for sl in (perm[:half], perm[half:]):
        pc = pcr.PointCloud.create(len(sl))
        pc.set_x_array(data["x"][sl])
        pc.set_y_array(data["y"][sl])
        pc.add_channel("value", pcr.DataType.Float32)
        pc.set_channel_array_f32("value", data["v"][sl])
        other.ingest(pc)                                              # host-resident: staged by the pipeline
    other.finalize()
    s2, c2, mn2, mx2 = _bands(other)
    assert other.last_scatter()["path"] == "direct"
    assert np.array_equal(c, c2, equal_nan=True)                      # integers: bit-exact
    assert np.array_equal(mn, mn2, equal_nan=True) and np.array_equal(mx, mx2, equal_nan=True)
    np.testing.assert_allclose(s, s2, rtol=1e-5, atol=1e-5)           # fp32 re-association only


@pytest.mark.parametrize("height", [16384, 8192], ids=["37504-tiles:two-level", "11008-tiles:one-level"])
def test_large_grid_paths_agree_at_full_size(height):
    """16384 x 16384 with four planes (37 504 Point tiles: the two-level sort) and 16384 x 8192 with Sum + Count (11 008 tiles, the
    window of a C5 shard at N = 2: one sort level since round 5, kMaxBins = 12 160) with 30 M points, no oracle: Count / Max / Min must be bit-identical
    to the direct-atomics path, Sum within float32 accumulation noise, counts conserved.
    Runs in a fresh interpreter because it generates its points with torch, which must be imported before pcr."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize_large_grid_worker.py")
    out = subprocess.run([sys.executable, worker, str(height)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "large grid paths agree" in out.stdout
