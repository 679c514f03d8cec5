import sys, time, numpy as np
sys.path.insert(0, "pointcloud-raster_amd/python")
import pcr
G, n = 4096, 50_000_000
rng = np.random.default_rng(42)
x = rng.uniform(2, G - 2, n); y = rng.uniform(2, G - 2, n); v = rng.uniform(0, 1, n).astype(np.float32)
d = rng.uniform(0, np.pi, n).astype(np.float32)
c = pcr.PointCloud.create(n); c.set_x_array(x); c.set_y_array(y)
c.add_channel("value", pcr.DataType.Float32); c.set_channel_array_f32("value", v)
c.add_channel("direction", pcr.DataType.Float32); c.set_channel_array_f32("direction", d)
dev = c.to_device()
def run(spec, name):
    cfg = pcr.PipelineConfig(); cfg.grid.bounds = pcr.BBox(0., 0., float(G), float(G)); cfg.grid.compute_dimensions()
    cfg.exec_mode = pcr.ExecutionMode.GPU; cfg.reductions = [spec]; cfg.result_location = pcr.MemoryLocation.Device
    cfg.gpu_pool_size_bytes = 40 * n
    for rep in range(2):
        p = pcr.Pipeline.create(cfg); p.profile_enable(True); p.ingest(dev); p.finalize()
        k = {a: round(b[1], 3) for a, b in p.profile_read(True).items()}
    print(name, k, flush=True)
T = pcr.ReductionType
for rt in (T.WeightedAverage, T.Sum, T.Count):
    s = pcr.gaussian_splat_spec("value", default_sigma=1.0, max_radius_cells=4.0); s.type = rt
    run(s, f"gauss1 {rt}")
for rt in (T.WeightedAverage, T.Sum, T.Count):
    s = pcr.line_splat_spec("value", direction_channel="direction", default_half_length=16.0, max_radius_cells=18.0); s.type = rt
    run(s, f"line16 {rt}")
