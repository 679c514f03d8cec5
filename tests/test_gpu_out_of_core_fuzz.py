"""Randomised differential test of the out-of-core pipeline (host/src/pipeline.cpp, Pipeline::Banded; replaces the reference's
TileManager LRU + spill, src/engine/tile_manager.cpp:76-138, 183-375): random grids whose height is not a multiple of the tile,
random sets of reductions over the three glyphs, a device budget that cuts the grid into one to several tile rows per band, a
host budget that parks all / some / none of the bands as `.pcrt` files, clouds that reach all or a few bands (or none), an
optional filter, and -- half of the time -- a checkpoint in the middle: save_state, a NEW pipeline (out of core or in core) that
resumes from it.  Every band must equal the IN-CORE pipeline's on the same ingests: bit for bit where no float atomic is involved
and both pipelines are bound to take the same path (Max, Min, Count of Points and Lines), to fp32 re-association otherwise (a
band's sub-pipeline may pick the direct path where the whole grid is binned).  Seeds 0..11 in the suite; PCR_OOC_FUZZ_SEEDS=a:b
soaks a range (or a,b,c: those seeds).  A failure names its seed."""
import os

import numpy as np
import pytest

import pcr
import pcr_oracle_py as O
from test_gpu_pipeline_api import cloud_from, config_for

pytestmark = pytest.mark.gpu

RT = {"Sum": pcr.ReductionType.Sum, "Max": pcr.ReductionType.Max, "Min": pcr.ReductionType.Min, "Average": pcr.ReductionType.Average,
      "WeightedAverage": pcr.ReductionType.WeightedAverage, "Count": pcr.ReductionType.Count}


def seeds():
    env = os.environ.get("PCR_OOC_FUZZ_SEEDS")
    if env and ":" in env:
        a, b = env.split(":")
        return list(range(int(a), int(b)))
    if env:
        return [int(v) for v in env.split(",")]
    return list(range(12))


def build(seed):
    rng = np.random.default_rng(77000 + seed)
    W, H = int(rng.integers(200, 700)), int(rng.integers(300, 1100))
    cs = float(rng.choice([0.5, 1.0, 2.0]))
    tile = (int(rng.choice([64, 128, 200, 4096])), int(rng.choice([64, 100, 128, 256])))
    og = O.make_grid((10.0, -20.0, 10.0 + W * cs, -20.0 + H * cs), cell=(cs, -cs), tile=tile)
    specs = []
    for _ in range(int(rng.integers(1, 5))):
        kind = str(rng.choice(["point", "point", "gauss", "line"]))
        ch = str(rng.choice(["a", "b"]))
        if kind == "point":
            specs.append(dict(kind=kind, ch=ch, rname=str(rng.choice(list(RT)))))
        elif kind == "gauss":
            specs.append(dict(kind=kind, ch=ch, rname=str(rng.choice(["Sum", "Average", "WeightedAverage", "Count"])),
                              sigma=float(rng.choice([0.8, 1.5])) * cs, maxr=4.0))
        else:
            specs.append(dict(kind=kind, ch=ch, rname=str(rng.choice(["Sum", "WeightedAverage", "Count"])), hl=float(rng.uniform(1.0, 6.0)) * cs))
    filt = (float(rng.integers(1, 4)),) if rng.uniform() < 0.3 else None
    clouds = []
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.integers(1, 60000))
        shape = str(rng.choice(["all", "strip", "cluster", "outside"]))
        if shape == "all":
            x, y = rng.uniform(og.min_x - cs, og.max_x + cs, n), rng.uniform(og.min_y - cs, og.max_y + cs, n)
        elif shape == "strip":                                   # a few rows: most bands are not reached
            y0 = rng.uniform(og.min_y, og.max_y)
            x, y = rng.uniform(og.min_x, og.max_x, n), rng.uniform(y0, min(og.max_y, y0 + 30 * cs), n)
        elif shape == "cluster":                                 # on a tile corner
            cx = og.min_x + og.tile_width * cs * int(rng.integers(0, max(1, W // og.tile_width) + 1))
            cy = og.max_y - og.tile_height * cs * int(rng.integers(0, max(1, H // og.tile_height) + 1))
            x, y = rng.normal(cx, 6 * cs, n), rng.normal(cy, 6 * cs, n)
        else:
            x, y = rng.uniform(og.max_x + cs, og.max_x + 50 * cs, n), rng.uniform(og.min_y, og.max_y, n)
        clouds.append(dict(x=x, y=y, a=rng.normal(5.0, 2.0, n).astype(np.float32), b=rng.uniform(-1.0, 1.0, n).astype(np.float32),
                           cls=rng.integers(0, 4, n).astype(np.float32), dir=rng.uniform(0, 6.3, n).astype(np.float32),
                           loc=str(rng.choice(["host", "device"]))))
    per_cell = 4 * sum(3 if s["rname"] in ("Average", "WeightedAverage") else 2 for s in specs)        # planes + band, roughly
    total = W * H * per_cell
    knobs = dict(gpu=int(total * float(rng.choice([0.04, 0.15, 0.4, 0.7]))), host=int(rng.choice([1, max(2, int(total * 0.3)), 1 << 30])),
                 checkpoint_after=int(rng.integers(0, len(clouds))) if rng.uniform() < 0.5 else None,
                 resume_in_core=bool(rng.uniform() < 0.4))
    return og, specs, filt, clouds, knobs


def reductions(specs):
    out = []
    for i, s in enumerate(specs):
        if s["kind"] == "point":
            r = pcr.ReductionSpec()
            r.value_channel = s["ch"]
        elif s["kind"] == "gauss":
            r = pcr.gaussian_splat_spec(s["ch"], default_sigma=s["sigma"], max_radius_cells=s["maxr"])
        else:
            r = pcr.line_splat_spec(s["ch"], direction_channel="dir", default_half_length=s["hl"], max_radius_cells=16.0)
        r.type = RT[s["rname"]]
        r.output_band_name = f"band{i}"
        out.append(r)
    return out


def config(og, specs, filt, **kw):
    cfg = config_for(og, reductions(specs), **kw)
    if filt:
        f = pcr.FilterSpec()
        f.add("cls", pcr.CompareOp.Less, filt[0])
        cfg.filter = f
    return cfg


def to_cloud(c):
    return cloud_from(c["x"], c["y"], {k: c[k] for k in ("a", "b", "cls", "dir")}, c["loc"])


@pytest.mark.parametrize("seed", seeds())
def test_out_of_core_equals_in_core(seed, tmp_path):
    og, specs, filt, clouds, knobs = build(seed)
    incore = pcr.Pipeline.create(config(og, specs, filt))
    assert incore is not None, pcr.pipeline_create_error()
    for c in clouds:
        incore.ingest(to_cloud(c))
    incore.finalize()
    want = [np.array(incore.result().band_array(i)) for i in range(len(specs))]

    work, ck = str(tmp_path / "work"), str(tmp_path / "ck")
    ooc = pcr.Pipeline.create(config(og, specs, filt, gpu_memory_budget=max(knobs["gpu"], 1 << 12), host_cache_budget=knobs["host"], state_dir=work))
    assert ooc is not None, f"seed {seed}: {pcr.pipeline_create_error()}"
    banded = ooc.out_of_core()
    processed = 0
    for k, c in enumerate(clouds):
        ooc.ingest(to_cloud(c))
        if knobs["checkpoint_after"] == k:
            ooc.save_state(ck)
            processed = ooc.stats().points_processed
            del ooc                                              # the spill directory goes with it
            assert not os.path.exists(work) or os.listdir(work) == [], f"seed {seed}: a spill outlived its pipeline"
            kw = dict(state_dir=ck, resume=True)
            if not knobs["resume_in_core"]:
                kw.update(gpu_memory_budget=max(knobs["gpu"], 1 << 12), host_cache_budget=knobs["host"])
            ooc = pcr.Pipeline.create(config(og, specs, filt, **kw))
            assert ooc is not None, f"seed {seed}: resume: {pcr.pipeline_create_error()}"
    ooc.finalize()
    if knobs["checkpoint_after"] is None:
        assert ooc.stats().points_processed == incore.stats().points_processed, f"seed {seed}: points_processed"
    else:
        assert processed + ooc.stats().points_processed == incore.stats().points_processed, f"seed {seed}: points_processed over the checkpoint"
    got = [np.array(ooc.result().band_array(i)) for i in range(len(specs))]
    assert_bands_match(f"seed {seed} (banded={banded}, knobs={knobs})", og, specs, filt, clouds, got, want)


def assert_bands_match(desc, og, specs, filt, clouds, got_bands, want_bands):
    """Bands of two pipelines of THIS library over the same ingests (tests/test_gpu_sharded_fuzz.py uses it too): bit for bit
    where no float atomic is involved (Max, Min, Count of Points and Lines), to fp32 re-association otherwise."""
    for i, s in enumerate(specs):
        what = f"{desc} band {i}: {s['kind']}/{s['rname']}"
        got, want = got_bands[i], want_bands[i]
        assert got.shape == want.shape == (og.height, og.width), what
        gn, wn = np.isnan(got), np.isnan(want)
        assert np.array_equal(gn, wn), f"{what}: NaN mask differs in {(gn != wn).sum()} cells"
        m = ~gn
        if s["rname"] in ("Max", "Min") or (s["rname"] == "Count" and s["kind"] != "gauss"):
            assert np.array_equal(got[m], want[m]), what
            continue
        # sums: the same contributions in another order (per band: another path, other tiles, other atomics) -- the error of a
        # cell scales with the MAGNITUDES that met there (channel b is +-1: its sums cancel), which the oracle supplies
        cat = {k: np.concatenate([c[k] for c in clouds]) for k in ("x", "y", "a", "b", "cls", "dir")}
        keep = cat["cls"] < filt[0] if filt else np.ones(len(cat["x"]), dtype=bool)
        x, y, v = cat["x"][keep], cat["y"][keep], cat[s["ch"]][keep]
        vmax = float(np.max(np.abs(v), initial=0.0))
        ogl, chans = None, {}
        if s["kind"] == "gauss":
            ogl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=s["sigma"], sigma_y=s["sigma"], max_radius=s["maxr"])
        elif s["kind"] == "line":
            ogl, chans = O.make_glyph(O.GLYPH_LINE, half_length=s["hl"], max_radius=16.0), dict(direction=cat["dir"][keep])
        if s["rname"] == "Sum":
            scale = np.maximum(np.nan_to_num(O.run(og, O.SUM, x, y, np.abs(v), glyph=ogl, wide=True, **chans).astype(np.float64))[m], 1e-2)
        elif s["rname"] == "Count":
            scale = np.maximum(np.abs(want[m]), 1e-2)
        else:
            scale = np.maximum(np.abs(want[m]), max(vmax, 1e-2))      # a mean's error scales with the values, not with the mean
        err = np.abs(got[m].astype(np.float64) - want[m])
        tol = 2e-5 * scale
        if s["kind"] == "gauss":
            # a band's sub-pipeline may take another Gaussian path than the whole grid does, and the paths may disagree about
            # a contribution that sits on the reference's `w < 1e-6f` cut-off (glyph_kernels.cu:166; tests/test_gpu_pipeline_fuzz.py):
            # two such contributions per cell are allowed, visible only where the cell's total weight is itself tiny
            if s["rname"] == "Count":
                tol = tol + 2e-6
            elif s["rname"] == "Sum":
                tol = tol + 2e-6 * vmax
            else:
                wsum = O.run(og, O.COUNT, x, y, v, glyph=ogl, wide=True).astype(np.float64)
                tol = tol + 2e-6 * (vmax + np.abs(want[m])) / np.maximum(np.nan_to_num(wsum[m]), 1e-6)
        assert (err <= tol).all(), f"{what}: max err / tol {np.max(err / tol):.3g}"
