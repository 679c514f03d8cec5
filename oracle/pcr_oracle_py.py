"""ctypes front-end for the CPU oracle (and, when built, the reference-compiled checker).

TEST INFRASTRUCTURE ONLY -- see oracle/pcr_oracle.h.  Importers allowed: tests/,
__graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

SUM, MAX, MIN, AVERAGE, WEIGHTED_AVERAGE, COUNT = 0, 1, 2, 3, 4, 5
GLYPH_POINT, GLYPH_LINE, GLYPH_GAUSSIAN = 0, 1, 2
RTYPE_NAMES = {SUM: "Sum", MAX: "Max", MIN: "Min", AVERAGE: "Average",
               WEIGHTED_AVERAGE: "WeightedAverage", COUNT: "Count"}


class Grid(C.Structure):
    _fields_ = [("min_x", C.c_double), ("min_y", C.c_double),
                ("max_x", C.c_double), ("max_y", C.c_double),
                ("cell_size_x", C.c_double), ("cell_size_y", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32),
                ("tile_width", C.c_int32), ("tile_height", C.c_int32)]


class Glyph(C.Structure):
    _fields_ = [("type", C.c_int32),
                ("default_direction", C.c_float), ("default_half_length", C.c_float),
                ("default_sigma_x", C.c_float), ("default_sigma_y", C.c_float),
                ("default_rotation", C.c_float), ("max_radius_cells", C.c_float)]


class Points(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("value", C.c_void_p),
                ("direction", C.c_void_p), ("half_length", C.c_void_p),
                ("sigma_x", C.c_void_p), ("sigma_y", C.c_void_p), ("rotation", C.c_void_p),
                ("n", C.c_uint64)]


def make_grid(bounds, cell=(1.0, -1.0), tile=(4096, 4096), dims=None):
    """bounds = (min_x, min_y, max_x, max_y).  dims=None -> compute_dimensions()."""
    g = Grid(bounds[0], bounds[1], bounds[2], bounds[3], cell[0], cell[1], 0, 0, tile[0], tile[1])
    if dims is None:
        lib().pcro_compute_dimensions(C.byref(g), None, None)
    else:
        g.width, g.height = dims
    return g


def make_glyph(type=GLYPH_POINT, direction=0.0, half_length=1.0, sigma_x=1.0, sigma_y=1.0,
               rotation=0.0, max_radius=32.0):
    return Glyph(type, direction, half_length, sigma_x, sigma_y, rotation, max_radius)


def _ptr(a, dtype):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data


def make_points(x, y, value, direction=None, half_length=None, sigma_x=None, sigma_y=None,
                rotation=None):
    """Returns (Points, keepalive)."""
    keep = []
    vals = []
    for a, dt in ((x, np.float64), (y, np.float64), (value, np.float32),
                  (direction, np.float32), (half_length, np.float32),
                  (sigma_x, np.float32), (sigma_y, np.float32), (rotation, np.float32)):
        arr, p = _ptr(a, dt)
        keep.append(arr)
        vals.append(p)
    n = 0 if keep[0] is None else keep[0].shape[0]
    return Points(*vals, n), keep


class Predicate(C.Structure):
    _fields_ = [("channel", C.c_void_p), ("op", C.c_int32), ("value", C.c_float),
                ("set", C.c_void_p), ("set_size", C.c_int32)]


CMP = {"Equal": 0, "NotEqual": 1, "Less": 2, "LessEqual": 3, "Greater": 4, "GreaterEqual": 5,
       "InSet": 6, "NotInSet": 7}


def filter_mask(n, predicates):
    """predicates: list of (channel array f32, op name, value or list of set values) -> (uint8 mask, kept)."""
    keep = []
    arr = (Predicate * max(len(predicates), 1))()
    for k, (ch, op, val) in enumerate(predicates):
        ch = np.ascontiguousarray(ch, dtype=np.float32)
        keep.append(ch)
        arr[k].channel = ch.ctypes.data
        arr[k].op = CMP[op]
        if op in ("InSet", "NotInSet"):
            st = np.ascontiguousarray(val, dtype=np.float32)
            keep.append(st)
            arr[k].set, arr[k].set_size, arr[k].value = st.ctypes.data, len(st), 0.0
        else:
            arr[k].set, arr[k].set_size, arr[k].value = None, 0, float(val)
    mask = np.zeros(n, dtype=np.uint8)
    L = lib()
    L.pcro_filter_mask.restype = C.c_uint64
    L.pcro_filter_mask.argtypes = [C.POINTER(Predicate), C.c_int, C.c_uint64, C.c_void_p]
    kept = L.pcro_filter_mask(arr, len(predicates), n, mask.ctypes.data)
    return mask, int(kept)


_lib = None
_ref = None


def build(ref=True):
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    if ref:
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "_build", "libpcr_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.pcro_last_error.restype = C.c_char_p
        L.pcro_create.restype = C.c_void_p
        L.pcro_create.argtypes = [C.POINTER(Grid), C.c_int, C.POINTER(Glyph), C.c_int]
        L.pcro_destroy.argtypes = [C.c_void_p]
        L.pcro_ingest.argtypes = [C.c_void_p, C.POINTER(Points)]
        L.pcro_finalize.argtypes = [C.c_void_p, C.c_void_p]
        L.pcro_tile_touched.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.pcro_points_valid.argtypes = [C.c_void_p]
        L.pcro_points_valid.restype = C.c_uint64
        L.pcro_world_to_cell.argtypes = [C.POINTER(Grid), C.c_double, C.c_double,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.pcro_tile_cell_range.argtypes = [C.POINTER(Grid), C.c_int32, C.c_int32] + [C.POINTER(C.c_int32)] * 4
        L.pcro_init_state.argtypes = [C.c_int, C.c_void_p, C.c_int64]
        L.pcro_accumulate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64]
        L.pcro_merge_state.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.pcro_finalize_state.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.pcro_accumulate_glyph.argtypes = [C.POINTER(Glyph), C.c_int, C.POINTER(Points), C.c_void_p,
                                            C.c_int64, C.POINTER(Grid)] + [C.c_int32] * 4
        _lib = L
    return _lib


def ref_lib():
    """The reference-compiled checker, or None when oracle/_ref was never built."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libpcr_ref.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.pcr_ref_last_error.restype = C.c_char_p
        L.pcr_ref_init_state.argtypes = [C.c_int, C.c_void_p, C.c_int64]
        L.pcr_ref_accumulate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64]
        L.pcr_ref_merge_state.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.pcr_ref_finalize_state.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.pcr_ref_accumulate_glyph.argtypes = [C.POINTER(Glyph), C.c_int, C.POINTER(Points), C.c_void_p,
                                               C.c_int64, C.POINTER(Grid)] + [C.c_int32] * 4
        _ref = L
    return _ref


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def _check(rc):
    if rc != 0:
        raise OracleError(rc, lib().pcro_last_error().decode())


def world_to_cell(g, wx, wy):
    c, r = C.c_int32(0), C.c_int32(0)
    ok = lib().pcro_world_to_cell(C.byref(g), wx, wy, C.byref(c), C.byref(r))
    return c.value, r.value, bool(ok)


def tile_cell_range(g, tile_row, tile_col):
    v = [C.c_int32(0) for _ in range(4)]
    lib().pcro_tile_cell_range(C.byref(g), tile_row, tile_col, *[C.byref(a) for a in v])
    return tuple(a.value for a in v)


class Reduction:
    """One ReductionSpec run through the oracle: ingest(...)* then finalize() -> (H, W) float32."""

    def __init__(self, grid, rtype, glyph=None, wide=False):
        self.grid = grid
        self.glyph = glyph if glyph is not None else make_glyph()
        self.h = lib().pcro_create(C.byref(grid), rtype, C.byref(self.glyph), int(wide))
        if not self.h:
            raise OracleError(1, lib().pcro_last_error().decode())

    def ingest(self, x, y, value, **glyph_channels):
        pts, keep = make_points(x, y, value, **glyph_channels)
        _check(lib().pcro_ingest(self.h, C.byref(pts)))
        del keep

    def finalize(self):
        out = np.empty((self.grid.height, self.grid.width), dtype=np.float32)
        _check(lib().pcro_finalize(self.h, out.ctypes.data))
        return out

    def tile_touched(self, tile_row, tile_col):
        return bool(lib().pcro_tile_touched(self.h, tile_row, tile_col))

    def points_valid(self):
        return int(lib().pcro_points_valid(self.h))

    def close(self):
        if self.h:
            lib().pcro_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run(grid, rtype, x, y, value, glyph=None, wide=False, **glyph_channels):
    r = Reduction(grid, rtype, glyph, wide)
    try:
        r.ingest(x, y, value, **glyph_channels)
        return r.finalize()
    finally:
        r.close()


_cpu = None


def cpu_pipeline_lib():
    """oracle/pcr_cpu_pipeline.cpp: the reference's CPU stages (assign / serial std::sort / per-update
    `omp critical` / finalize) around the oracle's arithmetic.  Timed by bench.py's cpu_baseline leg."""
    global _cpu
    if _cpu is None:
        path = os.path.join(_HERE, "_build", "libpcr_cpu_pipeline.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.pcro_cpu_pipeline_run.argtypes = [C.POINTER(Grid), C.c_int, C.POINTER(Glyph), C.POINTER(Points),
                                            C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        L.pcro_cpu_pipeline_max_threads.restype = C.c_int
        _cpu = L
    return _cpu


def cpu_pipeline_run(grid, rtype, x, y, value, glyph=None, threads=0, **glyph_channels):
    """-> (band (H, W) float32, {stage: seconds}).  threads <= 0: OpenMP default (all cores)."""
    L = cpu_pipeline_lib()
    pts, keep = make_points(x, y, value, **glyph_channels)
    band = np.empty((grid.height, grid.width), dtype=np.float32)
    st = (C.c_double * 5)()
    gl = glyph if glyph is not None else make_glyph()
    rc = L.pcro_cpu_pipeline_run(C.byref(grid), rtype, C.byref(gl), C.byref(pts), int(threads), band.ctypes.data, st)
    del keep
    if rc != 0:
        raise OracleError(rc, "pcro_cpu_pipeline_run failed")
    return band, dict(zip(("assign", "sort", "extract_batches", "accumulate", "finalize"), list(st)))
