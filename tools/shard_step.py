"""One rank's step of the C5 configuration (BASELINE configs[4]) on ONE GPU, without the collectives: rank R of W owns its row block
of the 16384^2 grid and the 1e9 / W points that fall into it; Point / Average and Gaussian sigma = 1 / Average.  The exchange is
mimicked by what it does to the pipeline (plane and flag pointers handed out: the finalize pass runs on its own).
Run on the GPU box: python tools/shard_step.py [W=8] [R=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
bench._imports()
from bench import pcr, make_specs
from pcr.distributed import row_block

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
G, N = 16384, 1_000_000_000 // W
r0, r1 = row_block(R, W, G, 1)


def device_cloud(n, seed):
    c = pcr.PointCloud.create(n, pcr.MemoryLocation.Device)
    c.resize(n)
    c.add_channel("value", pcr.DataType.Float32)
    ptr = c.device_ptrs()
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    x = torch.as_tensor(pcr.DeviceArrayView(ptr["x"], (n,), "<f8", owner=c), device="cuda")
    y = torch.as_tensor(pcr.DeviceArrayView(ptr["y"], (n,), "<f8", owner=c), device="cuda")
    x.uniform_(2.0, G - 2.0, generator=gen)
    y.uniform_(float(G - r1) + 0.01, float(G - r0) - 0.01, generator=gen)       # rows [r0, r1) of a north-up grid
    torch.as_tensor(pcr.DeviceArrayView(ptr["value"], (n,), "<f4", owner=c), device="cuda").uniform_(0.0, 1.0, generator=gen)
    return c


cloud = device_cloud(N, 42 + R)
for wl in ("C5_point", "C5_gauss1"):
    cfg = pcr.PipelineConfig()
    cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(G)); cfg.grid.cell_size_x, cfg.grid.cell_size_y = 1.0, -1.0
    cfg.grid.compute_dimensions(); cfg.exec_mode = pcr.ExecutionMode.GPU
    cfg.reductions = make_specs(wl); cfg.result_location = pcr.MemoryLocation.Device
    cfg.shard_row_begin, cfg.shard_row_end = r0, r1
    cfg.gpu_pool_size_bytes = 24 * N + (64 << 20)
    pipes = [pcr.Pipeline.create(cfg) for _ in range(6)]
    for i, p in enumerate(pipes):
        p.profile_enable(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p.ingest(cloud)
        p.state_planes(); p.tile_touched_ptr()                 # what the exchange does to the pipeline's bookkeeping
        p.finalize()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        k = {n: round(ms, 4) for n, (l, ms) in p.profile_read(True).items()}
        if i >= 2:
            print(f"{wl} rank {R}/{W} rows [{r0}, {r1}) {N / 1e6:.0f} M pts: {dt * 1e3:.3f} ms  valid {p.last_scatter()['points_valid']}  {k}", flush=True)
    del pipes
