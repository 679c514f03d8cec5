"""Throughput of the file formats either side of the hot path (run on the GPU box; files in $TMPDIR).
  PCRP 50 M points (x, y, value = 1.0 GB): write, read to Host, read to Device (pinned landing + one H2D copy)
  GeoTIFF 4096^2 x 3 bands (201 MB): write per compression, read back."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (shares libamdhip64 with pcr)
sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
import pcr

d = tempfile.mkdtemp(prefix="pcr_io_")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
rng = np.random.default_rng(0)
pc = pcr.PointCloud.create(n)
pc.set_x_array(rng.uniform(0, 4096, n))
pc.set_y_array(rng.uniform(0, 4096, n))
pc.add_channel("value", pcr.DataType.Float32)
pc.set_channel_array_f32("value", rng.uniform(0, 1, n).astype(np.float32))
gb = n * 20 / 1e9
p = os.path.join(d, "c.pcrp")
t = time.perf_counter(); pcr.write_point_cloud(p, pc); dt = time.perf_counter() - t
print(f"PCRP write  {gb:.2f} GB  {dt*1e3:8.1f} ms  {gb/dt:6.2f} GB/s")
for loc, name in ((pcr.MemoryLocation.Host, "Host"), (pcr.MemoryLocation.HostPinned, "HostPinned"), (pcr.MemoryLocation.Device, "Device")):
    for rep in range(2):
        t = time.perf_counter(); c = pcr.read_point_cloud(p, pcr.PointCloudFormat.Auto, loc); dt = time.perf_counter() - t
        del c
    print(f"PCRP read -> {name:10s} {dt*1e3:8.1f} ms  {gb/dt:6.2f} GB/s   (page cache warm)")
# file -> finalized grid, two ways (C2 reductions on 4096^2)
def pipeline():
    c = pcr.PipelineConfig()
    c.grid.bounds = pcr.BBox(0.0, 0.0, 4096.0, 4096.0); c.grid.cell_size_x, c.grid.cell_size_y = 1.0, -1.0
    c.grid.compute_dimensions(); c.exec_mode = pcr.ExecutionMode.GPU
    specs = []
    for t in (pcr.ReductionType.Sum, pcr.ReductionType.Count, pcr.ReductionType.Average):
        r = pcr.ReductionSpec(); r.value_channel, r.type = "value", t; specs.append(r)
    c.reductions = specs
    c.result_location = pcr.MemoryLocation.Device
    return pcr.Pipeline.create(c)
for rep in range(2):
    pp = pipeline()
    t = time.perf_counter()
    c = pcr.read_point_cloud(p, pcr.PointCloudFormat.Auto, pcr.MemoryLocation.Device); pp.ingest(c); pp.finalize()
    dt_whole = time.perf_counter() - t
    del c, pp
for chunk in (4 << 20, 16 << 20):
    for rep in range(2):
        pp = pipeline()
        t = time.perf_counter(); got = pp.ingest_file(p, chunk); pp.finalize(); dt = time.perf_counter() - t
        del pp
    print(f"file -> grid, ingest_file chunk {chunk>>20:3d} M pts: {dt*1e3:8.1f} ms  {n/dt/1e6:7.1f} Mpts/s   ({got} points)")
print(f"file -> grid, read whole to HBM then ingest: {dt_whole*1e3:8.1f} ms  {n/dt_whole/1e6:7.1f} Mpts/s")
W = H = 4096
bands = []
for nm in ("sum", "count", "avg"):
    b = pcr.BandDesc(); b.name = nm; bands.append(b)
g = pcr.Grid.create(W, H, bands)
cnt = rng.poisson(3.0, (H, W)).astype(np.float32)
g.set_band_array(0, (cnt * rng.uniform(0.4, 0.6, (H, W))).astype(np.float32))
g.set_band_array(1, cnt)
g.set_band_array(2, np.where(cnt > 0, rng.uniform(0.4, 0.6, (H, W)), np.nan).astype(np.float32))
cfg = pcr.GridConfig()
cfg.bounds = pcr.BBox(0, 0, W, H); cfg.cell_size_x, cfg.cell_size_y = 1.0, -1.0; cfg.compute_dimensions()
cfg.crs = pcr.CRS.from_epsg(32618)
mb = W * H * 3 * 4 / 1e6
for comp in ("NONE", "LZW", "DEFLATE"):
    o = pcr.GeoTiffOptions(); o.compress = comp
    q = os.path.join(d, f"o_{comp}.tif")
    t = time.perf_counter(); pcr.write_geotiff(q, g, cfg, o); dt = time.perf_counter() - t
    t2 = time.perf_counter(); a = pcr.read_geotiff_band(q, 2); dr = time.perf_counter() - t2
    print(f"GeoTIFF {comp:8s} write {dt*1e3:8.1f} ms {mb/dt:8.1f} MB/s  file {os.path.getsize(q)/1e6:7.1f} MB   read 1 band {dr*1e3:7.1f} ms")
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
