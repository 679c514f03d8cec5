"""Debug helper (GPU box): python tools/debug_seed.py SEED -- re-runs one mixed-pipeline fuzz case and prints where a band differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", os.path.join("pointcloud-raster_amd", "python")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import pcr
import pcr_oracle_py as O
import test_gpu_pipeline_fuzz as P

seed = int(sys.argv[1])
og, specs, filt, clouds = P.build(seed)
print("grid", og.width, og.height, "tile", og.tile_width, og.tile_height, "filter", filt)


def run(spec_ids, cloud_ids, device=None, path=0):
    cfg = pcr.PipelineConfig()
    cfg.grid.bounds = pcr.BBox(og.min_x, og.min_y, og.max_x, og.max_y)
    cfg.grid.cell_size_x, cfg.grid.cell_size_y = og.cell_size_x, og.cell_size_y
    cfg.grid.tile_width, cfg.grid.tile_height = og.tile_width, og.tile_height
    cfg.grid.compute_dimensions()
    cfg.exec_mode = pcr.ExecutionMode.GPU
    cfg.scatter_path = path
    rs = []
    for i in spec_ids:
        s = specs[i]
        if s["kind"] == "point":
            r = pcr.ReductionSpec(); r.value_channel = s["ch"]
        elif s["kind"] == "gauss":
            r = pcr.gaussian_splat_spec(s["ch"], default_sigma=s["sigma"], max_radius_cells=s["maxr"])
        else:
            r = pcr.line_splat_spec(s["ch"], direction_channel="dir", default_half_length=s["hl"], max_radius_cells=16.0)
        r.type = P.RT[s["rname"]][0]
        rs.append(r)
    cfg.reductions = rs
    if filt:
        f = pcr.FilterSpec(); f.add(filt[0], P.OPS[filt[1]], filt[2]); cfg.filter = f
    pipe = pcr.Pipeline.create(cfg)
    for k in cloud_ids:
        c = dict(clouds[k])
        if device is not None:
            c["device"] = device
        pipe.ingest(P.to_cloud(c))
    pipe.finalize()
    print("   last_scatter", pipe.last_scatter())
    return [np.array(pipe.result().band_array(b)) for b in range(len(spec_ids))]


def oracle(i, cloud_ids):
    s = specs[i]
    cat = {k: np.concatenate([clouds[c][k] for c in cloud_ids]) for k in ("x", "y", "a", "b", "cls", "dir")}
    chans = dict(direction=cat["dir"]) if s["kind"] == "line" else {}
    return O.run(og, P.RT[s["rname"]][1], cat["x"], cat["y"], cat[s["ch"]], glyph=s["ogl"], wide=True, **chans).astype(np.float64)


def report(tag, got, want):
    gn, wn = np.isnan(got), np.isnan(want)
    d = np.abs(np.nan_to_num(got.astype(np.float64)) - np.nan_to_num(want))
    idx = np.unravel_index(np.argmax(d), d.shape)
    print(f"{tag}: nan mismatches {(gn != wn).sum()}, max abs diff {d.max():.6g} at {idx}: got {got[idx]}, want {want[idx]}; cells off by > 1e-3: {(d > 1e-3).sum()}")


all_ids = list(range(len(clouds)))
for i, s in enumerate(specs):
    if s["kind"] == "gauss":
        print("spec", i, {k: v for k, v in s.items() if k != "ogl"})
        got = run(list(range(len(specs))), all_ids)[i]
        want = oracle(i, all_ids)
        cat = {k: np.concatenate([clouds[c][k] for c in all_ids]) for k in ("x", "y", "a", "b")}
        wsum = O.run(og, O.COUNT, cat["x"], cat["y"], cat[s["ch"]], glyph=s["ogl"], wide=True).astype(np.float64)
        mag = np.maximum(np.abs(want), np.nan_to_num(O.run(og, P.RT[s["rname"]][1], cat["x"], cat["y"], np.abs(cat[s["ch"]]), glyph=s["ogl"], wide=True)))
        both = ~np.isnan(got) & ~np.isnan(want)
        rel = np.zeros_like(want); rel[both] = np.abs(got[both] - want[both]) / np.maximum(1e-2, mag[both])
        idx = np.unravel_index(np.argmax(rel), rel.shape)
        print(f"  worst cell {idx}: got {got[idx]!r} want {want[idx]!r} rel {rel[idx]:.3g} total weight there {wsum[idx]:.6g} mag {mag[idx]:.4g}; cells over 1e-4: {(rel > 1e-4).sum()}")
        for pth in (1, 2):
            g2 = run([i], all_ids, path=pth)[0]
            print(f"  path {pth}: got {g2[idx]!r} rel {abs(g2[idx] - want[idx]) / max(1e-2, mag[idx]):.3g}")
        continue
    if s["kind"] != "line":
        continue
    print("spec", i, {k: v for k, v in s.items() if k != "ogl"})
    report("  full pipeline, all clouds", run(list(range(len(specs))), all_ids)[i], oracle(i, all_ids))
    report("  line alone, all clouds", run([i], all_ids)[0], oracle(i, all_ids))
    for k in all_ids:
        report(f"  line alone, cloud {k} (as generated: device={clouds[k]['device']})", run([i], [k])[0], oracle(i, [k]))
        report(f"  line alone, cloud {k}, direct path", run([i], [k], path=1)[0], oracle(i, [k]))
    report("  line alone, all clouds, all host", run([i], all_ids, device=False)[0], oracle(i, all_ids))
