"""Wall-clock breakdown of one C2 step (ingest / finalize), for diagnosing host-side overheads.
Run on the GPU box: python tools/step_breakdown.py [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import bench
bench._imports()                                   # bench imports torch / pcr lazily (its launcher branch must not)
from bench import pcr, ShardedPipeline, make_points, make_cloud, make_specs

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
G, n = 4096, 50_000_000
cfg = pcr.PipelineConfig()
cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(G))
cfg.grid.cell_size_x, cfg.grid.cell_size_y = 1.0, -1.0
cfg.grid.compute_dimensions()
cfg.exec_mode = pcr.ExecutionMode.GPU
cfg.reductions = make_specs(wl)
cfg.result_location = pcr.MemoryLocation.Device
cfg.gpu_pool_size_bytes = 16 * n + (64 << 20)
x, y, v, ch = make_points(wl, n, G, 0.0, float(G), seed=42)
cloud = make_cloud(x, y, v, ch).to_device()
t = time.perf_counter()
pipes = [ShardedPipeline(cfg, 0, 1, device_id=0) for _ in range(8)]
torch.cuda.synchronize()
print(f"create x8: {(time.perf_counter()-t)*1e3:.2f} ms")
for i, sp in enumerate(pipes):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sp.pipe.ingest(cloud)
    t1 = time.perf_counter()
    sp.pipe.synchronize()
    t2 = time.perf_counter()
    sp.finalize()
    t3 = time.perf_counter()
    print(f"step {i}: ingest call {1e3*(t1-t0):.3f}  ingest sync {1e3*(t2-t1):.3f}  finalize {1e3*(t3-t2):.3f} ms")
