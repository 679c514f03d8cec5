"""Seeded input recipes shared by the fixture generator (make_ref_vectors.py) and the tests.

A case describes ONE tile-state accumulation, i.e. one call of the reference's
accumulate_glyph() (src/engine/glyph_kernels.cu:571-604) or Accumulator::accumulate()
(src/engine/accumulator.cpp:33-59) on a batch whose points all have their centre cell in
the tile [col0, col0+tw) x [row0, row0+th) of the grid.
"""
import numpy as np

SUM, MAX, MIN, AVERAGE, WEIGHTED_AVERAGE, COUNT = 0, 1, 2, 3, 4, 5
POINT, LINE, GAUSSIAN = 0, 1, 2

# grid: bounds (min_x, min_y, max_x, max_y), cell (csx, csy); tile rect in cells.
_G96 = dict(bounds=(0.0, 0.0, 96.0, 96.0), cell=(1.0, -1.0), dims=(96, 96))
_G96_FULL = dict(col0=0, row0=0, tw=96, th=96)
_G_SUB = dict(col0=32, row0=16, tw=48, th=40)          # interior tile of the 96x96 grid -> clipping (Q4)
_G_HALF = dict(bounds=(100.0, -50.0, 148.0, -2.0), cell=(0.5, -0.5), dims=(96, 96))   # non-unit cells, offset origin

GLYPH_CASES = [
    dict(name="gauss_s1_wavg", glyph=GAUSSIAN, rtype=WEIGHTED_AVERAGE, grid=_G96, tile=_G96_FULL,
         n=3000, seed=1, sigma=(1.0, 1.0), max_radius=4.0),
    dict(name="gauss_s4_r12_avg", glyph=GAUSSIAN, rtype=AVERAGE, grid=_G96, tile=_G96_FULL,
         n=600, seed=2, sigma=(4.0, 4.0), max_radius=12.0),
    dict(name="gauss_s2_sum_clip", glyph=GAUSSIAN, rtype=SUM, grid=_G96, tile=_G_SUB,
         n=1500, seed=3, sigma=(2.0, 2.0), max_radius=32.0),
    dict(name="gauss_aniso_rot_count", glyph=GAUSSIAN, rtype=COUNT, grid=_G96, tile=_G96_FULL,
         n=800, seed=4, sigma=(3.0, 1.0), rotation=0.6, max_radius=32.0),
    dict(name="gauss_perpoint_channels", glyph=GAUSSIAN, rtype=WEIGHTED_AVERAGE, grid=_G96, tile=_G96_FULL,
         n=800, seed=5, sigma=(1.5, 1.5), max_radius=10.0,
         channels=("sigma_x", "sigma_y", "rotation")),
    dict(name="gauss_halfcell_grid", glyph=GAUSSIAN, rtype=WEIGHTED_AVERAGE, grid=_G_HALF, tile=_G96_FULL,
         n=1000, seed=6, sigma=(0.75, 1.25), max_radius=8.0),
    dict(name="line_hl16_wavg", glyph=LINE, rtype=WEIGHTED_AVERAGE, grid=_G96, tile=_G96_FULL,
         n=2000, seed=7, half_length=16.0, max_radius=18.0, channels=("direction",)),
    dict(name="line_hl4_sum_clip", glyph=LINE, rtype=SUM, grid=_G96, tile=_G_SUB,
         n=2000, seed=8, half_length=4.0, direction=0.5, max_radius=32.0),
    dict(name="line_perpoint_len_count", glyph=LINE, rtype=COUNT, grid=_G_HALF, tile=_G96_FULL,
         n=1500, seed=9, half_length=1.0, max_radius=6.0, channels=("direction", "half_length")),
    dict(name="line_axis_directions", glyph=LINE, rtype=COUNT, grid=_G96, tile=_G96_FULL,
         n=1200, seed=10, half_length=3.0, max_radius=32.0, channels=("direction",),
         axis_dirs=True),
]

POINT_CASES = [
    dict(name=f"point_{nm}", rtype=rt, tile_cells=48 * 40, n=20000, seed=20 + rt)
    for nm, rt in (("sum", SUM), ("max", MAX), ("min", MIN), ("avg", AVERAGE),
                   ("wavg", WEIGHTED_AVERAGE), ("count", COUNT))
]


def glyph_inputs(case):
    """Points whose centre cell lies inside the case's tile rect (as a TileBatch would hold)."""
    rng = np.random.default_rng(case["seed"])
    g, t = case["grid"], case["tile"]
    csx, csy = g["cell"]
    n = case["n"]
    # fractional cell coordinates uniformly inside the tile rect, then to world
    fcx = rng.uniform(t["col0"], t["col0"] + t["tw"], n)
    fcy = rng.uniform(t["row0"], t["row0"] + t["th"], n)
    x = g["bounds"][0] + fcx * csx
    y = g["bounds"][3] + fcy * csy
    value = rng.uniform(0.0, 1.0, n).astype(np.float32)
    ch = {}
    names = case.get("channels", ())
    if "direction" in names:
        if case.get("axis_dirs"):
            # exact multiples of pi/2 and pi/4 as float32: exercises cos(pi/2f) = -4.4e-8 jogs (Q7)
            ch["direction"] = (rng.integers(0, 8, n) * np.float32(np.pi / 4)).astype(np.float32)
        else:
            ch["direction"] = rng.uniform(0.0, np.pi, n).astype(np.float32)
    if "half_length" in names:
        ch["half_length"] = rng.uniform(0.2, 3.0, n).astype(np.float32)
    if "sigma_x" in names:
        sx = rng.uniform(0.5, 2.5, n).astype(np.float32)
        sx[::7] = 0.0          # <= 0 falls back to the default (glyph_kernels.cu:120-123)
        ch["sigma_x"] = sx
    if "sigma_y" in names:
        sy = rng.uniform(0.5, 2.5, n).astype(np.float32)
        sy[::11] = -1.0
        ch["sigma_y"] = sy
    if "rotation" in names:
        ch["rotation"] = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    return x, y, value, ch


def point_inputs(case):
    rng = np.random.default_rng(case["seed"])
    n, cells = case["n"], case["tile_cells"]
    ci = rng.integers(0, cells, n).astype(np.uint32)
    # leave a band of cells empty so NaN-on-empty is exercised
    ci[ci % 13 == 0] = 1
    v = rng.normal(0.0, 100.0, n).astype(np.float32)
    return ci, v
