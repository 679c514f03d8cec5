"""Adversarial inputs for the Point path: every point in ONE cell / one tile / one column; sorted input."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
bench._imports()                                   # bench imports torch / pcr lazily (its launcher branch must not)
from bench import pcr, make_cloud, make_specs

G, n = 4096, 50_000_000
cfg = pcr.PipelineConfig()
cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(G)); cfg.grid.cell_size_x, cfg.grid.cell_size_y = 1.0, -1.0
cfg.grid.compute_dimensions(); cfg.exec_mode = pcr.ExecutionMode.GPU
cfg.reductions = make_specs("C2"); cfg.result_location = pcr.MemoryLocation.Device
rng = np.random.default_rng(0)
v = rng.uniform(0, 1, n).astype(np.float32)
cases = {
    "one cell": (np.full(n, 1000.5), np.full(n, 2000.5)),
    "one 128x96 tile": (rng.uniform(1024, 1152, n), rng.uniform(960, 1056, n)),
    "one column": (np.full(n, 77.5), rng.uniform(0, G, n)),
    "sorted by y": (rng.uniform(0, G, n), np.sort(rng.uniform(0, G, n))),
    "uniform": (rng.uniform(0, G, n), rng.uniform(0, G, n)),
}
for name, (x, y) in cases.items():
    cloud = make_cloud(x, y, v, {}).to_device()
    ts = []
    for rep in range(3):
        p = pcr.Pipeline.create(cfg)
        torch.cuda.synchronize(); t = time.perf_counter()
        p.ingest(cloud); p.finalize()
        ts.append(time.perf_counter() - t)
        if rep == 2:
            cnt = torch.as_tensor(pcr.DeviceArrayView(p.result().band_device_ptr(1), (G, G), "<f4", owner=p), device="cuda")
            total = float(torch.nan_to_num(cnt).double().sum())
        del p
    print(f"{name:18s} {min(ts)*1e3:8.2f} ms  {n/min(ts)/1e9:6.1f} Gpts/s   count sum {total:.0f}")
    del cloud
