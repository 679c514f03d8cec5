"""The arithmetic behind round 4's Line records (scatter_binned_glyph.hip, LineStateMaker / k_tile_line_rec), checked on the
CPU against the reference's own walk.

The reference walks a segment with two error tests per step (src/engine/glyph_kernels.cu:252-278; restated in
oracle/pcr_oracle.c).  The tile kernel instead steps a Bresenham REMAINDER: the j-th visited cell is major = j,
minor = k_j = floor((2 j m + M - 1) / 2M), and clipping to a rectangle keeps a contiguous range of j that the scatter pass
solves in closed form.  This file restates both in Python and compares them cell for cell, clip rectangles included; the
HIP kernels themselves are compared with the oracle by the GPU suite (Count bit-exact)."""
import random


def reference_walk(x0, y0, x1, y1):
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx = 1 if x0 < x1 else -1
    sy = 1 if y0 < y1 else -1
    err, cx, cy, out = dx - dy, x0, y0, []
    while True:
        out.append((cx, cy))
        if cx == x1 and cy == y1:
            return out
        e2 = 2 * err
        if e2 > -dy:
            err -= dy
            cx += sx
        if e2 < dx:
            err += dx
            cy += sy


def first_j_with_k(k, M, m):
    num = 2 * M * k - M + 1
    return 0 if num <= 0 else (num + 2 * m - 1) // (2 * m)


def state_walk(x0, y0, x1, y1, clip):
    """LineStateMaker::make + the loop of k_tile_line_rec."""
    cx0, cx1, cy0, cy1 = clip
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    xmajor = dx >= dy
    M, m = (dx, dy) if xmajor else (dy, dx)
    sx = 1 if x0 < x1 else -1
    sy = 1 if y0 < y1 else -1
    js, je = 0, M
    if not (min(x0, x1) >= cx0 and max(x0, x1) < cx1 and min(y0, y1) >= cy0 and max(y0, y1) < cy1):
        u0, su, ulo, uhi = (x0, sx, cx0, cx1) if xmajor else (y0, sy, cy0, cy1)
        w0, sw, wlo, whi = (y0, sy, cy0, cy1) if xmajor else (x0, sx, cx0, cx1)
        js = max(js, ulo - u0 if su > 0 else u0 - (uhi - 1))
        je = min(je, uhi - 1 - u0 if su > 0 else u0 - ulo)
        ka = wlo - w0 if sw > 0 else w0 - (whi - 1)
        kb = whi - 1 - w0 if sw > 0 else w0 - wlo
        if m == 0:
            if ka > 0 or kb < 0:
                je = -1
        else:
            if ka > 0:
                js = max(js, first_j_with_k(ka, M, m))
            if kb < m:
                je = min(je, -1 if kb < 0 else first_j_with_k(kb + 1, M, m) - 1)
    if je < js:
        return []
    k = rem = 0
    if M > 0:
        N = 2 * js * m + M - 1
        k, rem = divmod(N, 2 * M)
    cx = x0 + sx * js if xmajor else x0 + sx * k
    cy = y0 + sy * k if xmajor else y0 + sy * js
    M2, m2, out = (2 * M if M > 0 else 1), 2 * m, []
    for _ in range(je - js + 1):
        out.append((cx, cy))
        rem += m2
        c = rem >= M2
        if c:
            rem -= M2
        if xmajor:
            cx += sx
            cy += sy if c else 0
        else:
            cy += sy
            cx += sx if c else 0
    return out


def test_remainder_walk_visits_the_reference_cells_for_every_small_segment():
    for x1 in range(-12, 13):
        for y1 in range(-12, 13):
            assert state_walk(0, 0, x1, y1, (-99, 99, -99, 99)) == reference_walk(0, 0, x1, y1), (x1, y1)


def test_clipping_is_a_j_range():
    rng = random.Random(1)
    for _ in range(60_000):
        x0, y0, x1, y1 = (rng.randint(-40, 40) for _ in range(4))
        clip = (rng.randint(-45, 0), rng.randint(0, 45), rng.randint(-45, 0), rng.randint(0, 45))
        want = [c for c in reference_walk(x0, y0, x1, y1) if clip[0] <= c[0] < clip[1] and clip[2] <= c[1] < clip[3]]
        assert state_walk(x0, y0, x1, y1, clip) == want, (x0, y0, x1, y1, clip)
