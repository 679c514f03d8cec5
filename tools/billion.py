"""C5 at single-GPU scale: 1 G uniform points generated in HBM, 16384^2 grid, Sum+Count+Average in ONE ingest
(two-level sort, 20 GB of sort scratch), then the same in 8 ingests of 125 M points.  Checks the point count."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
bench._imports()                                   # bench imports torch / pcr lazily (its launcher branch must not)
from bench import pcr, make_specs

G = 16384
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
cfg = pcr.PipelineConfig()
cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(G)); cfg.grid.cell_size_x, cfg.grid.cell_size_y = 1.0, -1.0
cfg.grid.compute_dimensions(); cfg.exec_mode = pcr.ExecutionMode.GPU
cfg.reductions = make_specs("C2"); cfg.result_location = pcr.MemoryLocation.Device


def device_cloud(n, seed):
    c = pcr.PointCloud.create(n, pcr.MemoryLocation.Device)
    c.resize(n)
    c.add_channel("value", pcr.DataType.Float32)
    ptr = c.device_ptrs()
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    for name, dt in (("x", "<f8"), ("y", "<f8")):
        t = torch.as_tensor(pcr.DeviceArrayView(ptr[name], (n,), dt, owner=c), device="cuda")
        t.uniform_(2.0, G - 2.0, generator=gen)
    torch.as_tensor(pcr.DeviceArrayView(ptr["value"], (n,), "<f4", owner=c), device="cuda").uniform_(0.0, 1.0, generator=gen)
    return c


def count_sum(p):
    cnt = torch.as_tensor(pcr.DeviceArrayView(p.result().band_device_ptr(1), (G, G), "<f4", owner=p), device="cuda")
    return float(torch.nan_to_num(cnt).double().sum())

cloud = device_cloud(N, 1)
torch.cuda.synchronize()
for rep in range(2):
    p = pcr.Pipeline.create(cfg)
    torch.cuda.synchronize(); t = time.perf_counter()
    p.ingest(cloud); p.finalize()
    dt = time.perf_counter() - t
    info = p.last_scatter()
    print(f"one ingest of {N/1e6:.0f} M points: {dt*1e3:8.1f} ms  {N/dt/1e9:6.2f} Gpts/s  path {info['path']} bins {info['num_bins']}  count sum {count_sum(p):.0f}", flush=True)
    del p
del cloud
chunk = N // 8
clouds = [device_cloud(chunk, 10 + i) for i in range(2)]
p = pcr.Pipeline.create(cfg)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(8):
    p.ingest(clouds[i % 2])
p.finalize()
dt = time.perf_counter() - t
print(f"8 ingests of {chunk/1e6:.0f} M points:  {dt*1e3:8.1f} ms  {8*chunk/dt/1e9:6.2f} Gpts/s  count sum {count_sum(p):.0f}")
