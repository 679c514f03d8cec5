#!/usr/bin/env python3
"""What an out-of-core pipeline costs: ms per ingest and per band at 16384^2, Point /
Average, against the same cloud through the in-core pipeline.  Runs ON THE GPU BOX:
    python tools/ooc_cost.py [points] > gpurun_out/ooc_cost.json"""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
os.environ.setdefault("PCR_REQUIRE_GPU_ENGINE", "1")
import numpy as np   # noqa: E402
import pcr           # noqa: E402

G = 16384
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000


def cfg(**kw):
    c = pcr.PipelineConfig()
    c.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(G))
    c.grid.cell_size_x, c.grid.cell_size_y = 1.0, -1.0
    c.grid.compute_dimensions()
    c.exec_mode = pcr.ExecutionMode.GPU
    r = pcr.ReductionSpec()
    r.value_channel, r.type = "value", pcr.ReductionType.Average
    c.reductions = [r]
    for k, v in kw.items():
        setattr(c, k, v)
    return c


rng = np.random.default_rng(42)
cloud = pcr.PointCloud.create(n)
cloud.set_x_array(rng.uniform(2, G - 2, n))
cloud.set_y_array(rng.uniform(2, G - 2, n))
cloud.add_channel("value", pcr.DataType.Float32)
cloud.set_channel_array_f32("value", rng.uniform(0, 1, n).astype(np.float32))
dev = cloud.to_device()
del cloud


def run(label, the_cloud=None, **kw):
    p = pcr.Pipeline.create(cfg(**kw))
    assert p is not None, pcr.pipeline_create_error()
    times = []
    for _ in range(3):                       # three ingests of the same cloud into one pipeline
        t0 = time.perf_counter()
        p.ingest(the_cloud if the_cloud is not None else dev)
        p.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    p.finalize()
    fin = (time.perf_counter() - t0) * 1e3
    band = np.array(p.result().band_array(0))
    out = {"what": label, "out_of_core": bool(p.out_of_core()), "ingest_ms": [round(t, 1) for t in times], "finalize_ms": round(fin, 1),
           "checksum": float(np.nansum(band.astype(np.float64)))}
    spill = p.spill_dir() if p.out_of_core() else ""
    if spill:
        files = [os.path.join(d, f) for d, _, fs in os.walk(spill) for f in fs]
        out["spill_files"] = len(files)
        out["spill_MB"] = round(sum(os.path.getsize(f) for f in files) / 1e6, 1)
    return out


res = {"grid": f"{G}x{G}", "points": n, "state": "Average: 2 planes x 1.07 GB + one 1.07 GB band", "runs": []}
res["runs"].append(run("in core"))
tmp = tempfile.mkdtemp(prefix="pcr_ooc_")
try:
    # 1.2 GB budget: one 4096-row tile row (805 MB of planes + band) per band -> 4 bands
    res["runs"].append(run("out of core, 4 bands parked in host memory", gpu_memory_budget=1200 << 20, host_cache_budget=64 << 30, state_dir=tmp))
    res["runs"].append(run("out of core, 4 bands spilled to disk (.pcrt tiles)", gpu_memory_budget=1200 << 20, host_cache_budget=1, state_dir=tmp))
    # a survey tile: the same number of points inside the top-left 4096 x 4096 cells -- one band of the four is reached
    m = n // 4
    c2 = pcr.PointCloud.create(m)
    c2.set_x_array(rng.uniform(2, 4094, m))
    c2.set_y_array(rng.uniform(G - 4094, G - 2, m))
    c2.add_channel("value", pcr.DataType.Float32)
    c2.set_channel_array_f32("value", rng.uniform(0, 1, m).astype(np.float32))
    d2 = c2.to_device()
    res["runs"].append(run(f"in core, {m} points inside one 4096^2 corner", the_cloud=d2))
    res["runs"].append(run(f"out of core (host-parked), {m} points inside one 4096^2 corner: one band reached", the_cloud=d2,
                           gpu_memory_budget=1200 << 20, host_cache_budget=64 << 30, state_dir=tmp))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
for i, r in enumerate(res["runs"]):
    if not r["out_of_core"]:
        base = r
        continue
    r["bands"] = 4
    r["ms_per_ingest_per_band"] = round(sum(r["ingest_ms"][1:]) / 2 / 4, 1)
    r["same_result_as_in_core"] = r["checksum"] == base["checksum"]
print(json.dumps(res, indent=1))
