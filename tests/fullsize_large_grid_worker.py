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
    # argv[1] = grid height: 16384 (21 888 Point tiles: the two-level sort) or 8192 (11 008 tiles: ONE level since round 5 --
    # the window of a C5 shard at N = 2)
    G2, n = 16384, 30_000_000
    H = int(sys.argv[1]) if len(sys.argv) > 1 else G2
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    cloud = pcr.PointCloud.create(n, pcr.MemoryLocation.Device)
    cloud.resize(n)
    cloud.add_channel("value", pcr.DataType.Float32)
    ptr = cloud.device_ptrs()
    for name in ("x", "y"):
        torch.as_tensor(pcr.DeviceArrayView(ptr[name], (n,), "<f8", owner=cloud), device="cuda").uniform_(-3.0, (G2 if name == "x" else H) + 3.0, generator=gen)
    torch.as_tensor(pcr.DeviceArrayView(ptr["value"], (n,), "<f4", owner=cloud), device="cuda").normal_(0.0, 5.0, generator=gen)
    out = {}
    for path in (1, 0):                                   # direct, auto (= binned, two-level)
        cfg = pcr.PipelineConfig()
        cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G2), float(H))
        cfg.grid.compute_dimensions()
        cfg.exec_mode = pcr.ExecutionMode.GPU
        # four planes: LDS tiles of 128 x 56 (37 504 of them at 16384^2); Sum + Count alone: 128 x 96 (11 008 at 16384 x 8192)
        kinds = [pcr.ReductionType.Count, pcr.ReductionType.Max, pcr.ReductionType.Min, pcr.ReductionType.Sum] if H == G2 else \
                [pcr.ReductionType.Count, pcr.ReductionType.Count, pcr.ReductionType.Count, pcr.ReductionType.Sum]
        cfg.reductions = [_spec(k) for k in kinds]
        cfg.scatter_path = path
        cfg.result_location = pcr.MemoryLocation.Device
        p = pcr.Pipeline.create(cfg)
        assert p is not None, pcr.pipeline_create_error()
        p.profile_enable(True)
        p.ingest(cloud)
        p.finalize()
        kernels = set(p.profile_read(True))
        info = p.last_scatter()
        assert info["path"] == ("direct" if path == 1 else "binned"), info
        if path == 0:
            assert ("k_sub_scatter" in kernels) == (H == 16384), kernels          # the second sort level only beyond kMaxBins tiles
            assert info["num_bins"] == (37504 if H == 16384 else 11008 if H == 8192 else info["num_bins"]), info
        bands = [torch.as_tensor(pcr.DeviceArrayView(p.result().band_device_ptr(b), (H, G2), "<f4", owner=p), device="cuda").clone()
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
