"""ctypes view of the C-ABI (include/pcr_hip.h) of lib/libpcr_hip.so.

This is what a foreign-language binding of the engine looks like; the
parity tests drive the HIP path through it.  There is no fallback: if the shared library is
missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libpcr_hip.so"))

SUM, MAX, MIN, AVERAGE, WEIGHTED_AVERAGE, COUNT = 0, 1, 2, 3, 4, 5
GLYPH_POINT, GLYPH_LINE, GLYPH_GAUSSIAN = 0, 1, 2
PLANE_SUM, PLANE_WGT, PLANE_MAX, PLANE_MIN = 1, 2, 4, 8
PATH_AUTO, PATH_DIRECT, PATH_BINNED = 0, 1, 2


class Grid(C.Structure):
    _fields_ = [("min_x", C.c_double), ("min_y", C.c_double), ("max_x", C.c_double), ("max_y", C.c_double),
                ("cell_size_x", C.c_double), ("cell_size_y", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32),
                ("tile_width", C.c_int32), ("tile_height", C.c_int32),
                ("own_row0", C.c_int32), ("own_row1", C.c_int32),
                ("state_row0", C.c_int32), ("state_rows", C.c_int32)]


class Planes(C.Structure):
    _fields_ = [("d_sum", C.c_void_p), ("d_wgt", C.c_void_p), ("d_max", C.c_void_p), ("d_min", C.c_void_p)]


class Glyph(C.Structure):
    _fields_ = [("type", C.c_int32),
                ("default_direction", C.c_float), ("default_half_length", C.c_float),
                ("default_sigma_x", C.c_float), ("default_sigma_y", C.c_float),
                ("default_rotation", C.c_float), ("max_radius_cells", C.c_float),
                ("d_direction", C.c_void_p), ("d_half_length", C.c_void_p),
                ("d_sigma_x", C.c_void_p), ("d_sigma_y", C.c_void_p), ("d_rotation", C.c_void_p)]


class Predicate(C.Structure):
    _fields_ = [("d_channel", C.c_void_p), ("op", C.c_int32), ("value", C.c_float), ("set_size", C.c_int32),
                ("set", C.c_float * 16)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint32), ("total_ms", C.c_double)]


class ScatterStats(C.Structure):
    _fields_ = [("points_in", C.c_uint64), ("points_valid", C.c_uint64), ("path", C.c_int32),
                ("lds_tile_w", C.c_int32), ("lds_tile_h", C.c_int32), ("lds_apron", C.c_int32),
                ("num_bins", C.c_int32), ("scatter_chunk", C.c_int32), ("reserved_", C.c_int32)]


# every symbol include/pcr_hip.h declares: name -> argtypes (restype is int unless noted)
_VP, _SZ, _I64, _U64, _U32 = C.c_void_p, C.c_size_t, C.c_int64, C.c_uint64, C.c_uint32
class HaloPlane(C.Structure):
    _fields_ = [("d_plane", C.c_void_p), ("kind", C.c_uint32), ("reserved_", C.c_uint32)]


class HaloGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "state_row0", "state_rows", "own_row0", "own_row1", "halo", "nplanes",
                                         "kinds", "valid", "reserved_")]


class XferGeom(C.Structure):
    _fields_ = [("narrays", C.c_int32), ("elem_bytes", C.c_int32 * 8), ("root", C.c_int32), ("valid", C.c_int32),
                ("reserved_", C.c_int32), ("recv_capacity", C.c_uint64), ("send_counts", C.c_uint64 * 64)]


SYMBOLS = {
    "pcr_hip_last_error": None,
    "pcr_hip_abi_version": [],
    "pcr_hip_device_count": [C.POINTER(C.c_int)],
    "pcr_hip_set_device": [C.c_int],
    "pcr_hip_get_device": [C.POINTER(C.c_int)],
    "pcr_hip_device_name": [C.c_int, C.c_char_p, _SZ],
    "pcr_hip_mem_info": [C.POINTER(_SZ), C.POINTER(_SZ)],
    "pcr_hip_device_synchronize": [],
    "pcr_hip_stream_create": [C.POINTER(_VP)],
    "pcr_hip_stream_destroy": [_VP],
    "pcr_hip_stream_synchronize": [_VP],
    "pcr_hip_event_create": [C.POINTER(_VP)],
    "pcr_hip_event_destroy": [_VP],
    "pcr_hip_event_record": [_VP, _VP],
    "pcr_hip_event_elapsed_ms": [_VP, _VP, C.POINTER(C.c_float)],
    "pcr_hip_malloc": [C.POINTER(_VP), _SZ],
    "pcr_hip_free": [_VP],
    "pcr_hip_host_alloc": [C.POINTER(_VP), _SZ],
    "pcr_hip_host_free": [_VP],
    "pcr_hip_memcpy_h2d": [_VP, _VP, _SZ, _VP],
    "pcr_hip_memcpy_d2h": [_VP, _VP, _SZ, _VP],
    "pcr_hip_memcpy_d2d": [_VP, _VP, _SZ, _VP],
    "pcr_hip_copy_kernel": [_VP, _VP, _SZ, C.c_int, _VP],
    "pcr_hip_memset": [_VP, C.c_int, _SZ, _VP],
    "pcr_hip_arena_create": [C.POINTER(_VP), _SZ],
    "pcr_hip_arena_destroy": [_VP],
    "pcr_hip_arena_alloc": [_VP, _SZ, C.POINTER(_VP)],
    "pcr_hip_arena_reset": [_VP],
    "pcr_hip_arena_stats": [_VP, C.POINTER(_SZ), C.POINTER(_SZ), C.POINTER(_SZ)],
    "pcr_hip_device_scratch_stats": [C.c_int, C.POINTER(_SZ), C.POINTER(_SZ), C.POINTER(_U64), C.POINTER(_U64)],
    "pcr_hip_route_count": [C.POINTER(Grid), C.POINTER(C.c_int32), C.c_int, _VP, _VP, _VP, _U64, _VP, _VP, _VP],
    "pcr_hip_route_scatter": [_VP, _U64, C.c_int, _VP, C.c_int, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(C.c_int32), _VP],
    "pcr_hip_absmax_f32": [_VP, _U64, C.POINTER(C.c_float), _VP],
    "pcr_hip_absmax_f32_masked": [_VP, _VP, _U64, _VP, C.POINTER(C.c_float), _VP],
    "pcr_hip_signed_max_f32_masked": [_VP, _VP, _U64, _VP, C.POINTER(C.c_float), C.POINTER(C.c_float), _VP],
    "pcr_hip_comm_available": [],
    "pcr_hip_comm_unique_id": [_VP],
    "pcr_hip_comm_create": [C.POINTER(_VP), _VP, C.c_int, C.c_int, C.c_int],
    "pcr_hip_comm_destroy": [_VP],
    "pcr_hip_comm_rank": [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "pcr_hip_comm_halo_reduce": [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP],
    "pcr_hip_comm_halo_plan": [C.POINTER(HaloGeom), C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4,
    "pcr_hip_comm_agree_max_i32": [_VP, C.POINTER(C.c_int32), _VP],
    "pcr_hip_comm_allreduce_max_u32": [_VP, _VP, C.c_int, _VP],
    "pcr_hip_comm_allreduce_sum_f64": [_VP, _VP, C.c_int, _VP],
    "pcr_hip_comm_stats": [_VP, C.POINTER(_U64), C.POINTER(_U64)],
    "pcr_hip_comm_alltoall_counts": [_VP, C.POINTER(_U64), C.POINTER(_U64), _VP],
    "pcr_hip_comm_alltoallv": [_VP, C.c_int, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(C.c_int32), C.POINTER(_U64), _U64,
                               C.POINTER(_U64), _VP],
    "pcr_hip_comm_gatherv": [_VP, C.c_int, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(C.c_int32), _U64, _U64, C.POINTER(_U64),
                             C.c_int, _VP],
    "pcr_hip_comm_xfer_plan": [C.POINTER(XferGeom), C.c_int, C.c_int, C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U64)],
    "pcr_hip_state_floats": [C.c_int, C.POINTER(C.c_int)],
    "pcr_hip_plane_fill": [_VP, C.c_float, _I64, _VP],
    "pcr_hip_state_init": [C.c_int, _VP, _I64, _VP],
    "pcr_hip_state_merge": [C.c_int, _VP, _VP, _I64, _VP],
    "pcr_hip_plane_merge": [_U32, _VP, _VP, _I64, _VP],
    "pcr_hip_finalize": [C.c_int, C.POINTER(Grid), C.POINTER(Planes), _VP, _VP, _VP],
    "pcr_hip_finalize_group": [C.POINTER(Grid), C.POINTER(Planes), _VP, C.c_int, C.POINTER(C.c_int), C.POINTER(_VP), _VP],
    "pcr_hip_touched_union": [_VP, _VP, C.c_int32, _VP, C.c_int32, _VP],
    "pcr_hip_touched_union_owned": [_VP, _VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _VP, C.c_int32, _VP],
    "pcr_hip_finalize_group_unless": [C.POINTER(Grid), C.POINTER(Planes), _VP, C.c_int, C.POINTER(C.c_int), C.POINTER(_VP), _VP, _VP],
    "pcr_hip_engine_create": [C.POINTER(_VP), C.POINTER(Grid), _SZ, _VP],
    "pcr_hip_engine_destroy": [_VP],
    "pcr_hip_engine_set_path": [_VP, C.c_int],
    "pcr_hip_engine_planes_fresh": [_VP, C.c_int],
    "pcr_hip_engine_finalize_with_scatter": [_VP, C.c_int, C.POINTER(C.c_int), C.POINTER(_VP), _VP],
    "pcr_hip_engine_finalize_taken": [_VP],
    "pcr_hip_engine_stats": [_VP, C.POINTER(ScatterStats)],
    "pcr_hip_engine_tile_touched": [_VP, C.POINTER(_VP), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "pcr_hip_filter_mask": [C.POINTER(Predicate), C.c_int, _U64, _VP, _VP, _VP],
    "pcr_hip_engine_set_point_mask": [_VP, _VP],
    "pcr_hip_engine_profile_enable": [_VP, C.c_int],
    "pcr_hip_engine_profile_only": [_VP, C.c_char_p],
    "pcr_hip_engine_profile_read": [_VP, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int), C.c_int],
    "pcr_hip_scatter_point": [_VP, _U32, C.POINTER(Planes), _VP, _VP, _VP, _U64],
    "pcr_hip_scatter_glyph": [_VP, C.POINTER(Glyph), _U32, C.POINTER(Planes), _VP, _VP, _VP, _U64],
}

_lib = None


class PcrHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def lib():
    """Loads libpcr_hip.so (raises if it is not built) and types every exported symbol."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python pointcloud-raster_amd/build.py`")
        L = C.CDLL(LIB_PATH)
        for name, args in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            if name == "pcr_hip_last_error":
                fn.restype = C.c_char_p
                fn.argtypes = []
            else:
                fn.restype = C.c_int
                fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise PcrHipError(rc, lib().pcr_hip_last_error().decode())


def device_count():
    n = C.c_int(0)
    check(lib().pcr_hip_device_count(C.byref(n)))
    return n.value


class DeviceBuffer:
    """A device allocation made through the C-ABI."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(lib().pcr_hip_malloc(C.byref(self.ptr), max(self.nbytes, 1)))

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        if a.nbytes:
            check(lib().pcr_hip_memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes, None))
            check(lib().pcr_hip_stream_synchronize(None))
        b.dtype, b.shape = a.dtype, a.shape
        return b

    def to_numpy(self, dtype=None, shape=None):
        dtype = np.dtype(dtype or self.dtype)
        shape = shape if shape is not None else getattr(self, "shape", (self.nbytes // dtype.itemsize,))
        out = np.empty(shape, dtype=dtype)
        if out.nbytes:
            check(lib().pcr_hip_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None))
            check(lib().pcr_hip_stream_synchronize(None))
        return out

    def free(self):
        if self.ptr:
            lib().pcr_hip_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def make_grid(bounds, cell=(1.0, -1.0), dims=None, tile=(4096, 4096), own_rows=None, halo=0):
    min_x, min_y, max_x, max_y = [float(b) for b in bounds]
    if dims is None:
        import math
        dims = (int(math.ceil((max_x - min_x) / abs(cell[0]))), int(math.ceil((max_y - min_y) / abs(cell[1]))))
    w, h = dims
    r0, r1 = own_rows if own_rows is not None else (0, h)
    s0, s1 = max(0, r0 - halo), min(h, r1 + halo)
    return Grid(min_x, min_y, max_x, max_y, cell[0], cell[1], w, h, tile[0], tile[1], r0, r1, s0, s1 - s0)


class ReductionRun:
    """One accumulation group on the device, driven purely through the C-ABI:
    planes + engine; scatter(...)*; finalize(rtype) -> numpy (own rows x width)."""

    def __init__(self, grid, plane_mask, path=PATH_AUTO, engine=None):
        L = lib()
        self.grid = grid
        self.mask = plane_mask
        self.cells = grid.state_rows * grid.width
        self.bufs = {}
        self.planes = Planes()
        for bit, name, ident in ((PLANE_SUM, "d_sum", 0.0), (PLANE_WGT, "d_wgt", 0.0),
                                 (PLANE_MAX, "d_max", -3.4028234663852886e38),
                                 (PLANE_MIN, "d_min", 3.4028234663852886e38)):
            if plane_mask & bit:
                b = DeviceBuffer(self.cells * 4)
                check(L.pcr_hip_plane_fill(b.ptr, ident, self.cells, None))
                self.bufs[name] = b
                setattr(self.planes, name, b.ptr.value)
        self.own_engine = engine is None
        if engine is None:
            self.engine = C.c_void_p()
            check(L.pcr_hip_engine_create(C.byref(self.engine), C.byref(grid), 0, None))
        else:
            self.engine = engine
        check(L.pcr_hip_engine_set_path(self.engine, path))
        self._keep = []

    def scatter(self, x, y, value, glyph=None, mask=None, **channels):
        L = lib()
        mask = self.mask if mask is None else mask
        dx = DeviceBuffer.from_numpy(np.asarray(x, dtype=np.float64))
        dy = DeviceBuffer.from_numpy(np.asarray(y, dtype=np.float64))
        dv = DeviceBuffer.from_numpy(np.asarray(value, dtype=np.float32))
        n = dx.shape[0]
        keep = [dx, dy, dv]
        if glyph is None:
            check(L.pcr_hip_scatter_point(self.engine, mask, C.byref(self.planes), dx.ptr, dy.ptr, dv.ptr, n))
        else:
            g = Glyph(glyph["type"], glyph.get("direction", 0.0), glyph.get("half_length", 1.0),
                      glyph.get("sigma_x", 1.0), glyph.get("sigma_y", 1.0), glyph.get("rotation", 0.0),
                      glyph.get("max_radius", 32.0), None, None, None, None, None)
            for name in ("direction", "half_length", "sigma_x", "sigma_y", "rotation"):
                if channels.get(name) is not None:
                    b = DeviceBuffer.from_numpy(np.asarray(channels[name], dtype=np.float32))
                    keep.append(b)
                    setattr(g, "d_" + name, b.ptr.value)
            check(L.pcr_hip_scatter_glyph(self.engine, C.byref(g), mask, C.byref(self.planes),
                                          dx.ptr, dy.ptr, dv.ptr, n))
        check(L.pcr_hip_stream_synchronize(None))
        del keep

    def stats(self):
        st = ScatterStats()
        check(lib().pcr_hip_engine_stats(self.engine, C.byref(st)))
        return st

    def touched(self):
        L = lib()
        p, tx, ty = C.c_void_p(), C.c_int32(0), C.c_int32(0)
        check(L.pcr_hip_engine_tile_touched(self.engine, C.byref(p), C.byref(tx), C.byref(ty)))
        out = np.empty((ty.value, tx.value), dtype=np.uint32)
        check(L.pcr_hip_memcpy_d2h(out.ctypes.data, p, out.nbytes, None))
        check(L.pcr_hip_stream_synchronize(None))
        return out, p

    def plane(self, name):
        return self.bufs[name].to_numpy(np.float32, (self.grid.state_rows, self.grid.width))

    def finalize(self, rtype, use_touched=True):
        L = lib()
        rows = self.grid.own_row1 - self.grid.own_row0
        out = DeviceBuffer(rows * self.grid.width * 4)
        tp = None
        if use_touched:
            _, tp = self.touched()
        check(L.pcr_hip_finalize(rtype, C.byref(self.grid), C.byref(self.planes), tp, out.ptr, None))
        check(L.pcr_hip_stream_synchronize(None))
        return out.to_numpy(np.float32, (rows, self.grid.width))

    def close(self):
        if self.own_engine and self.engine:
            lib().pcr_hip_engine_destroy(self.engine)
            self.engine = C.c_void_p()
        for b in self.bufs.values():
            b.free()
        self.bufs = {}
