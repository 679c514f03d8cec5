"""pcr -- Point Cloud Reduction, MI355X (gfx950) engine.

Drop-in for the reference `pcr` package (python/pcr/__init__.py): same classes, enums and
helper functions, backed by hand-written HIP kernels through the C-ABI of include/pcr_hip.h.
ExecutionMode.GPU / Auto / Hybrid run the HIP engine; ExecutionMode.CPU -- and the cases in which the
reference falls back to its CPU mode, after the reference's own Warning / Info line -- run the host
engine (`Pipeline.engine()` says which; PCR_REQUIRE_GPU_ENGINE=1 forbids every fallback).  The module
itself does not import without libpcr_hip.so.
"""
import os as _os

__version__ = "0.1.0+mi355x"

_lib = _os.path.normpath(_os.path.join(_os.path.dirname(__file__), "..", "..", "lib", "libpcr_hip.so"))
if not _os.path.exists(_lib):
    raise ImportError(f"pcr: {_lib} is not built -- run `python pointcloud-raster_amd/build.py` "
                      "(hipcc --offload-arch=gfx950). There is no fallback implementation.")



def _share_hip_runtime_with_torch():
    """One HIP runtime per process, whatever the import order.

    libpcr_hip.so needs `libamdhip64.so.7`; PyTorch-ROCm bundles its own copy under torch/lib with the same
    SONAME.  The dynamic loader keeps whichever copy is loaded FIRST for everybody, so `import pcr; import torch`
    used to run torch on the system runtime and `import torch; import pcr` the engine on torch's.  When torch is
    installed its copy is loaded here, before the engine: both orders end up on the same runtime, and device
    pointers, streams and events can be shared between the two (pcr.distributed relies on that)."""
    import ctypes
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                                     # torch's runtime is already the process's runtime
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return                                     # no torch: the system ROCm runtime
    lib_dir = _os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = _os.path.join(lib_dir, name)
        if _os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError as exc:                 # loud: a half-shared runtime is the trap this function removes
                raise ImportError(f"pcr: cannot preload torch's HIP runtime ({path}): {exc}") from exc


def hip_runtime_paths():
    """Paths of every libamdhip64 mapped into this process (diagnostic: more than one is a bug)."""
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    paths.add(line.split()[-1])
    except OSError:
        pass
    return sorted(paths)


_share_hip_runtime_with_torch()

from ._pcr import (  # noqa: E402,F401
    BBox, BandDesc, CRS, ChannelDesc, CompareOp, DataType, ExecutionMode, FilterPredicate, FilterSpec,
    GeoTiffOptions, GlyphSpec, GlyphType, Grid, GridConfig, MemoryLocation, NativeShardedPipeline, NoDataPolicy, Pipeline,
    PipelineConfig, PointCloud, PointCloudFormat, PointCloudInfo, PointCloudReader, ProgressInfo,
    ReductionSpec, ReductionType, Status, StatusCode, TileIndex,
    device_count, device_name, pipeline_create_error,
    read_geotiff_info, read_point_cloud, read_point_cloud_info, write_geotiff, write_point_cloud,
    read_tile_state, tile_state_filename, write_tile_state,
    read_geotiff_band, read_geotiff_band_names, TiledGeoTiffWriter,
)


def _splat_spec(value_channel, glyph_type, max_radius_cells, output_band_name):
    spec = ReductionSpec()
    spec.value_channel = value_channel
    spec.type = ReductionType.WeightedAverage
    spec.glyph.type = glyph_type
    spec.glyph.max_radius_cells = max_radius_cells
    if output_band_name:
        spec.output_band_name = output_band_name
    return spec


def gaussian_splat_spec(value_channel, sigma_x_channel="", sigma_y_channel="", rotation_channel="",
                        default_sigma=1.0, default_sigma_x=None, default_sigma_y=None,
                        default_rotation=0.0, max_radius_cells=32.0, output_band_name=None):
    """ReductionSpec (WeightedAverage) that paints every point as a Gaussian footprint.

    Sigmas are in world units; `default_sigma` seeds both axes unless `default_sigma_x` /
    `default_sigma_y` override it; per-point channels (Float32) override the defaults where
    positive.  The footprint is clamped to `max_radius_cells` cells in each direction.
    """
    spec = _splat_spec(value_channel, GlyphType.Gaussian, max_radius_cells, output_band_name)
    g = spec.glyph
    g.sigma_x_channel, g.sigma_y_channel, g.rotation_channel = sigma_x_channel, sigma_y_channel, rotation_channel
    g.default_sigma_x = default_sigma if default_sigma_x is None else default_sigma_x
    g.default_sigma_y = default_sigma if default_sigma_y is None else default_sigma_y
    g.default_rotation = default_rotation
    spec.glyph = g
    return spec


def line_splat_spec(value_channel, direction_channel="", half_length_channel="",
                    default_direction=0.0, default_half_length=1.0, max_radius_cells=32.0,
                    output_band_name=None):
    """ReductionSpec (WeightedAverage) that paints every point as a one-cell-wide Bresenham
    segment of length 2*half_length (world units) along `direction` (radians, 0 = +X)."""
    spec = _splat_spec(value_channel, GlyphType.Line, max_radius_cells, output_band_name)
    g = spec.glyph
    g.direction_channel, g.half_length_channel = direction_channel, half_length_channel
    g.default_direction, g.default_half_length = default_direction, default_half_length
    spec.glyph = g
    return spec


class DeviceArrayView:
    """Zero-copy handle on device memory owned by a pcr object, consumable by anything that
    understands `__cuda_array_interface__` (e.g. `torch.as_tensor(view, device="cuda")`)."""

    def __init__(self, ptr, shape, typestr, owner=None):
        self._owner = owner
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape), "typestr": typestr,
            "data": (int(ptr), False), "version": 3, "strides": None,
        }


__all__ = [
    "DataType", "ReductionType", "MemoryLocation", "ExecutionMode", "StatusCode", "CompareOp",
    "PointCloudFormat", "GlyphType",
    "BBox", "CRS", "NoDataPolicy", "TileIndex", "Status", "ChannelDesc", "BandDesc",
    "GridConfig", "Grid", "PointCloud", "FilterPredicate", "FilterSpec",
    "GlyphSpec", "ReductionSpec", "PipelineConfig", "ProgressInfo", "Pipeline", "NativeShardedPipeline",
    "gaussian_splat_spec", "line_splat_spec",
    "GeoTiffOptions", "write_geotiff", "read_geotiff_info",
    "PointCloudInfo", "read_point_cloud", "write_point_cloud", "read_point_cloud_info", "PointCloudReader",
    "DeviceArrayView", "hip_runtime_paths", "device_count", "device_name", "pipeline_create_error",
    "read_tile_state", "write_tile_state", "tile_state_filename",
    "read_geotiff_band", "read_geotiff_band_names", "TiledGeoTiffWriter",
]
