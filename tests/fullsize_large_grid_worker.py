"""Worker of tests/test_gpu_fullsize_properties.py::test_large_grid_paths_agree_at_full_size (own process: torch first)."""
import os
import sys

import torch  # noqa: F401  before pcr: one shared HIP runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
import pcr  # noqa: E402


def _spec(t, ch="value"):
    r = pcr.ReductionSpec()
    r.value_channel, r.type = ch, t
    return r


def main():
    G2, n = 16384, 30_000_000
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    cloud = pcr.PointCloud.create(n, pcr.MemoryLocation.Device)
    cloud.resize(n)
    cloud.add_channel("value", pcr.DataType.Float32)
    ptr = cloud.device_ptrs()
    for name in ("x", "y"):
        torch.as_tensor(pcr.DeviceArrayView(ptr[name], (n,), "<f8", owner=cloud), device="cuda").uniform_(-3.0, G2 + 3.0, generator=gen)
    torch.as_tensor(pcr.DeviceArrayView(ptr["value"], (n,), "<f4", owner=cloud), device="cuda").normal_(0.0, 5.0, generator=gen)
    out = {}
    for path in (1, 0):                                   # direct, auto (= binned, two-level)
        cfg = pcr.PipelineConfig()
        cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G2), float(G2))
        cfg.grid.compute_dimensions()
        cfg.exec_mode = pcr.ExecutionMode.GPU
        cfg.reductions = [_spec(pcr.ReductionType.Count), _spec(pcr.ReductionType.Max), _spec(pcr.ReductionType.Min),
                          _spec(pcr.ReductionType.Sum)]
        cfg.scatter_path = path
        cfg.result_location = pcr.MemoryLocation.Device
        p = pcr.Pipeline.create(cfg)
        assert p is not None, pcr.pipeline_create_error()
        p.ingest(cloud)
        p.finalize()
        info = p.last_scatter()
        assert info["path"] == ("direct" if path == 1 else "binned"), info
        if path == 0:
            assert info["num_bins"] > 8064, info
        bands = [torch.as_tensor(pcr.DeviceArrayView(p.result().band_device_ptr(b), (G2, G2), "<f4", owner=p), device="cuda").clone()
                 for b in range(4)]
        out[path] = (bands, info["points_valid"])
        del p
    (c1, mx1, mn1, s1), valid1 = out[1]
    (c0, mx0, mn0, s0), valid0 = out[0]
    assert valid0 == valid1 and float(torch.nan_to_num(c0).double().sum()) == valid0
    for a, b in ((c0, c1), (mx0, mx1), (mn0, mn1)):
        assert torch.equal(torch.isnan(a), torch.isnan(b))
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    assert torch.allclose(torch.nan_to_num(s0), torch.nan_to_num(s1), rtol=1e-5, atol=1e-3)
    print("large grid paths agree:", valid0, "valid points")


if __name__ == "__main__":
    main()
