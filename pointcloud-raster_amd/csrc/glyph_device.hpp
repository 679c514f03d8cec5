// glyph_device.hpp -- in-register glyph footprints (Gaussian, Bresenham line), shared by the
// direct (global atomics) and the binned (LDS tile) scatter kernels through a Sink:
//     sink.add(row, col, v_times_w, w)        row/col are GLOBAL cell coordinates, already clipped
//
// Arithmetic follows the reference's CPU glyph code, NOT its CUDA kernels
// (src/engine/glyph_kernels.cu:79-281): f64 fractional cell position via multiplication
// by 1/cell_size, f32 sub-cell offsets and weights, footprint clipped to the reference tile
// of the centre cell (quirk Q4), corner sampling (Q5), signed sigma_y (Q6), f64 line
// end points rounded half away from zero (Q7).
#pragma once

#include "libm_sincosf.hpp"

#include "common.hpp"

namespace pcrhip {

// Per-point set-up shared by both glyphs: validity, centre cell, clip rectangle.
struct PointGeom {
    double fcx, fcy;      // fractional cell coordinates (glyph_kernels.cu:110-111, 219-220)
    int col, row;         // centre cell from world_to_cell (decides the tile), clamped
    int cx0, cx1, cy0, cy1;   // clip rectangle [cx0,cx1) x [cy0,cy1): centre tile ∩ state window
    bool valid;
};

__device__ __forceinline__ PointGeom point_geom(const GridDev& g, double wx, double wy) {
    PointGeom p;
    p.valid = world_to_cell(g, wx, wy, p.col, p.row);
    p.valid = p.valid && p.row >= g.own_r0 && p.row < g.own_r1;
    p.fcx = (wx - g.min_x) * g.inv_csx;
    p.fcy = (wy - g.max_y) * g.inv_csy;
    int tcx = p.valid ? fast_div(p.col, g.tw) : 0, tcy = p.valid ? fast_div(p.row, g.th) : 0;
    p.cx0 = tcx * g.tw;
    p.cx1 = min(p.cx0 + g.tw, g.W);
    p.cy0 = max(tcy * g.th, g.st_r0);
    p.cy1 = min(min(tcy * g.th + g.th, g.H), g.st_r0 + g.st_rows);
    return p;
}

// cosf / sinf as the reference's host libm returns them -- glibc's own sincosf algorithm, restated in libm_sincosf.hpp and
// compared with the system's libm bit for bit (tests/test_libm_sincosf.py): a last-bit difference moves a Line's rounded
// end point by a whole cell now and then.  (Rounds 1-4 took (float)cos((double)a) for it; glibc's float routines are not
// correctly rounded, and 2.7 % of random arguments differ -- the round-5 soak found the segment that showed it.)
__device__ __forceinline__ void sincos_like_libm(float a, float& s, float& c) { libm::sincosf(a, s, c); }

// Per-point channel values of a glyph (only the ones whose GlyphDev pointer is set are meaningful):
// Gaussian: c0 = sigma_x, c1 = sigma_y, c2 = rotation; Line: c0 = direction, c1 = half_length.
struct GlyphChan {
    float c0, c1, c2;
};

__device__ __forceinline__ GlyphChan load_chan(const GlyphDev& gl, uint64_t i) {
    GlyphChan c{0.f, 0.f, 0.f};
    if (gl.type == PCR_HIP_GLYPH_GAUSSIAN) {
        if (gl.sigma_x) c.c0 = gl.sigma_x[i];
        if (gl.sigma_y) c.c1 = gl.sigma_y[i];
        if (gl.rotation) c.c2 = gl.rotation[i];
    } else {
        if (gl.direction) c.c0 = gl.direction[i];
        if (gl.half_length) c.c1 = gl.half_length[i];
    }
    return c;
}

// ---- Gaussian ------------------------------------------------------------------
struct GaussParams {
    float val, sub_cx, sub_cy, sx, sy, cos_r, sin_r;
    int icx, icy, r;
    int cx0, cx1, cy0, cy1;
};

// glyph_kernels.cu:105-145 for the lane's own point.
__device__ __forceinline__ GaussParams gauss_params(const GridDev& g, const GlyphDev& gl,
                                                    const PointGeom& pg, float val, const GlyphChan& ch) {
    GaussParams q;
    q.val = val;
    double flx = floor(pg.fcx), fly = floor(pg.fcy);
    q.sub_cx = (float)(pg.fcx - flx);
    q.sub_cy = (float)(pg.fcy - fly);
    float sxw = gl.def_sigma_x, syw = gl.def_sigma_y;
    if (gl.sigma_x) { float t = ch.c0; if (t > 0.0f) sxw = t; }
    if (gl.sigma_y) { float t = ch.c1; if (t > 0.0f) syw = t; }
    q.sx = sxw * (float)g.inv_csx;
    q.sy = syw * (float)g.inv_csy;
    float rot = gl.rotation ? ch.c2 : gl.def_rotation;
    sincos_like_libm(-rot, q.sin_r, q.cos_r);
    float R = fminf(3.0f * fmaxf(q.sx, q.sy), gl.max_radius);
    // a NaN/huge radius cannot run away: the clip rectangle bounds the loop below
    q.r = min((int)ceilf(R), 1 << 20);
    q.icx = (int)flx;
    q.icy = (int)fly;
    q.cx0 = pg.cx0; q.cx1 = pg.cx1; q.cy0 = pg.cy0; q.cy1 = pg.cy1;
    return q;
}

// Lane `src`'s parameters, made wave-uniform (v_readlane into SGPRs; src must be uniform).
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ int lane_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

__device__ __forceinline__ GaussParams lane_bcast(const GaussParams& q, int src) {
    GaussParams u;
    u.val = lane_bcast(q.val, src); u.sub_cx = lane_bcast(q.sub_cx, src); u.sub_cy = lane_bcast(q.sub_cy, src);
    u.sx = lane_bcast(q.sx, src); u.sy = lane_bcast(q.sy, src);
    u.cos_r = lane_bcast(q.cos_r, src); u.sin_r = lane_bcast(q.sin_r, src);
    u.icx = lane_bcast(q.icx, src); u.icy = lane_bcast(q.icy, src); u.r = lane_bcast(q.r, src);
    u.cx0 = lane_bcast(q.cx0, src); u.cx1 = lane_bcast(q.cx1, src);
    u.cy0 = lane_bcast(q.cy0, src); u.cy1 = lane_bcast(q.cy1, src);
    return u;
}

// Weight of offset (dx, dy): glyph_kernels.cu:157-166.
__device__ __forceinline__ float gauss_weight(const GaussParams& q, int dx, int dy) {
    float rdx = (float)dx - q.sub_cx;
    float rdy = (float)dy - q.sub_cy;
    float rxr = rdx * q.cos_r + rdy * (-q.sin_r);
    float ryr = rdx * q.sin_r + rdy * q.cos_r;
    float a = rxr / q.sx, b = ryr / q.sy;
    return expf(-0.5f * (a * a + b * b));
}

// One WAVE paints one point: the (2r+1)^2 window clipped to the clip rectangle is
// enumerated row-major, 64 cells per step, so consecutive lanes hit consecutive cells
// of a row (contiguous atomics).  q must be wave-uniform.
template <typename Sink>
__device__ __forceinline__ void gauss_splat_generic(const GaussParams& q, int lane, Sink& sink,
                                                    int x0, int y0, int wdt, int hgt) {
    int total = wdt * hgt;
    float inv_w = 1.0f / (float)wdt;
    const bool small = total < (1 << 18);                 // float row index exact below 2^21 cells
    for (int idx = lane; idx < total; idx += 64) {
        int ry = small ? (int)(((float)idx + 0.5f) * inv_w) : idx / wdt;
        int rx = idx - ry * wdt;
        int gx = x0 + rx, gy = y0 + ry;
        float w = gauss_weight(q, gx - q.icx, gy - q.icy);
        if (w < 1e-6f) continue;                          // glyph_kernels.cu:166
        sink.add(gy, gx, q.val * w, w);
    }
}

// One LANE paints one point: for footprints of a few dozen cells the per-point set-up of the
// wave-cooperative form costs more than the cells themselves.  Same arithmetic as the reference
// loop (glyph_kernels.cu:145-176); the row term is hoisted out of the column loop when the
// footprint is axis-aligned (exact: rxr == rdx, ryr == rdy there).
template <typename Sink>
__device__ __forceinline__ void gauss_splat_lane(const GaussParams& q, Sink& sink) {
    int x0 = max(q.icx - q.r, q.cx0), x1 = min(q.icx + q.r + 1, q.cx1);
    int y0 = max(q.icy - q.r, q.cy0), y1 = min(q.icy + q.r + 1, q.cy1);
    const bool axis_aligned = (q.cos_r == 1.0f) && (q.sin_r == 0.0f);
    for (int gy = y0; gy < y1; ++gy) {
        const float rdy = (float)(gy - q.icy) - q.sub_cy;
        const float by = rdy / q.sy;
        const float b2 = by * by;
        for (int gx = x0; gx < x1; ++gx) {
            float w;
            if (axis_aligned) {
                const float rdx = (float)(gx - q.icx) - q.sub_cx;
                const float a = rdx / q.sx;
                w = expf(-0.5f * (a * a + b2));
            } else {
                w = gauss_weight(q, gx - q.icx, gy - q.icy);
            }
            if (w < 1e-6f) continue;
            sink.add(gy, gx, q.val * w, w);
        }
    }
}

// One LANE paints one point whose radius R is known at compile time (default-sigma, unrotated glyphs: every
// point of the launch has the same r).  The weight is separable there, w = exp(-a^2/2) * exp(-b^2/2): 2(2R+1)
// exponentials and divisions per point instead of (2R+1)^2, and the cell loop is fully unrolled (a product of
// two correctly rounded factors is within 2 ulp of the reference's single expf; the 1e-6 cut-off is applied
// to it).  This kernel was VALU-bound on expf + division before.
template <int R, typename Sink>
__device__ __forceinline__ void gauss_splat_fixed(const GaussParams& q, Sink& sink) {
    float ex[2 * R + 1], ey[2 * R + 1];
#pragma unroll
    for (int j = 0; j <= 2 * R; ++j) {
        const float a = ((float)(j - R) - q.sub_cx) / q.sx;
        const float b = ((float)(j - R) - q.sub_cy) / q.sy;
        ex[j] = expf(-0.5f * (a * a));
        ey[j] = expf(-0.5f * (b * b));
    }
#pragma unroll
    for (int i = 0; i <= 2 * R; ++i) {
        const int gy = q.icy - R + i;
        if (gy < q.cy0 || gy >= q.cy1) continue;
#pragma unroll
        for (int j = 0; j <= 2 * R; ++j) {
            const int gx = q.icx - R + j;
            const float w = ex[j] * ey[i];
            if (gx < q.cx0 || gx >= q.cx1 || w < 1e-6f) continue;
            sink.add(gy, gx, q.val * w, w);
        }
    }
}

// One WAVE paints one point (q wave-uniform).
//
// Axis-aligned footprints (rotation 0: cos == 1, sin == 0, the default) are evaluated without any
// per-cell division: with no rotation rxr == rdx and ryr == rdy exactly, so (rxr/sx)^2 depends
// only on the column and (ryr/sy)^2 only on the row.  Lanes are laid out as G rows x wdt columns
// (G = 64 / wdt); each lane keeps its column's a^2 for the whole point (one true division per
// lane per point), the row terms b^2 are computed once per 64-row block (one division per lane)
// and fetched per step with ds_bpermute.  The weight is then the reference's own expression
// exp(-0.5f * (a*a + b*b)) with identically rounded operands.
// Rotated or very wide (> 192 columns after clipping) footprints use the generic loop.
template <typename Sink>
__device__ __forceinline__ void gauss_splat_wave(const GaussParams& q, int lane, Sink& sink) {
    // intersect the window with the clip rectangle first: no lane iterates outside it
    int x0 = max(q.icx - q.r, q.cx0), x1 = min(q.icx + q.r + 1, q.cx1);
    int y0 = max(q.icy - q.r, q.cy0), y1 = min(q.icy + q.r + 1, q.cy1);
    int wdt = x1 - x0, hgt = y1 - y0;
    if (wdt <= 0 || hgt <= 0) return;
    const bool axis_aligned = (q.cos_r == 1.0f) && (q.sin_r == 0.0f);
    if (!axis_aligned || wdt > 192) {
        gauss_splat_generic(q, lane, sink, x0, y0, wdt, hgt);
        return;
    }
    const bool narrow = wdt <= 64;
    const int G = narrow ? 64 / wdt : 1;                               // rows per step
    // g = lane / wdt, c = lane % wdt for lane < 64 (exact: lane * (ceil(2^16/wdt)*wdt - 2^16) < 2^16)
    const int magic = (65536 + wdt - 1) / wdt;
    const int g = narrow ? (lane * magic) >> 16 : 0;
    const int c_in = narrow ? lane - g * wdt : lane;
    const bool lane_used = narrow ? (g < G) : true;
    const int ncb = narrow ? 1 : (wdt + 63) >> 6;
    for (int cb = 0; cb < ncb; ++cb) {
        const int c = c_in + (cb << 6);
        const bool col_ok = lane_used && c < wdt;
        const int gx = x0 + c;
        const float rdx = (float)(gx - q.icx) - q.sub_cx;
        const float a = rdx / q.sx;
        const float a2 = a * a;
        for (int rb = 0; rb < hgt; rb += 64) {
            // row table: lane l holds (rdy/sy)^2 of row rb + l
            const float rdy = (float)(y0 + rb + lane - q.icy) - q.sub_cy;
            const float bt = rdy / q.sy;
            const float b2t = bt * bt;
            const int rows_here = min(64, hgt - rb);
            for (int k = 0; k < rows_here; k += G) {
                const int rloc = k + g;
                const float b2 = __builtin_bit_cast(
                    float, __builtin_amdgcn_ds_bpermute((rloc & 63) << 2, __builtin_bit_cast(int, b2t)));
                const float w = expf(-0.5f * (a2 + b2));
                if (col_ok && rloc < rows_here && !(w < 1e-6f))
                    sink.add(y0 + rb + rloc, gx, q.val * w, w);
            }
        }
    }
}

// ---- Line ------------------------------------------------------------------------
struct LineParams {
    float val;
    int ix0, iy0, ix1, iy1;
    int cx0, cx1, cy0, cy1;
};

// glyph_kernels.cu:213-250.
__device__ __forceinline__ LineParams line_params(const GridDev& g, const GlyphDev& gl,
                                                  const PointGeom& pg, float val, const GlyphChan& ch) {
    LineParams q;
    q.val = val;
    float direction = gl.direction ? ch.c0 : gl.def_direction;
    float half_len = gl.half_length ? ch.c1 : gl.def_half_length;
    float hx = half_len * (float)g.inv_csx;
    float hy = half_len * (float)g.inv_csy;
    hx = fminf(hx, gl.max_radius);                 // std::min(h, cap): hy < 0 is never capped (Q7)
    hy = fminf(hy, gl.max_radius);
    float sd, cd;
    sincos_like_libm(direction, sd, cd);
    float px = hx * cd, py = hy * sd;              // f32 products, then f64 sums (:240-243)
    double x0 = pg.fcx - (double)px, y0 = pg.fcy - (double)py;
    double x1 = pg.fcx + (double)px, y1 = pg.fcy + (double)py;
    q.ix0 = (int)round(x0); q.iy0 = (int)round(y0);
    q.ix1 = (int)round(x1); q.iy1 = (int)round(y1);
    q.cx0 = pg.cx0; q.cx1 = pg.cx1; q.cy0 = pg.cy0; q.cy1 = pg.cy1;
    return q;
}

// sin and cos of |a| <= 64 in single precision, ~35 instructions: three-term Cody-Waite reduction by pi/2, the cephes
// minimax polynomials on [-pi/4, pi/4].  Relative error <= 1.3e-7 (~2 ulp) on that range (checked against numpy over
// 4e6 random arguments per range, incl. the neighbourhoods of the zeros); larger arguments lose the reduction.
__device__ __forceinline__ void sincos_f32_small(float a, float& s, float& c) {
    const float k = rintf(a * 0.63661977236758134f);
    float r = fmaf(-k, 1.5703125f, a);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188e-8f, r);
    const float z = r * r;
    const float sp = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    const float sr = fmaf(sp * z, r, r);
    const float cp = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    const float cr = fmaf(cp * z, z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k;
    const float s1 = (q & 1) ? cr : sr, c1 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s1 : s1;
    c = ((q + 1) & 2) ? -c1 : c1;
}
// The end points of line_params() with a SINGLE-precision sincos instead of the f64 one -- for the binning pass, where the
// f64 sincos is most of the vector work.  The reference's cosf is glibc's (sincos_like_libm, within 0.56 ulp), so this
// cd may differ from it by a couple of ulp, which moves an end point by |h| * 6e-7 cells at most: the ROUNDED end points
// are the reference's unless a coordinate sits that close to a half-integer.  `ambiguous` says so (margin: 6e-7 relative
// on the product -- 5 ulp -- plus 1e-7 cells), and for directions beyond +-64 rad; the caller then takes line_params().
// round() rounds halves away from zero, floor(v + 1/2) differs from it on negative halves only -- which are ambiguous by
// the margin test anyway.  A NaN anywhere is ambiguous (every comparison below fails the other way round).  (The rounding
// itself in single precision -- the centre converted to f32 -- makes 0.8 % of the points ambiguous at column 4096 and the
// pass slower again: measured, kept in f64.)
__device__ __forceinline__ LineParams line_params_fast(const GridDev& g, const GlyphDev& gl, const PointGeom& pg, float val,
                                                       const GlyphChan& ch, bool& ambiguous) {
    LineParams q;
    q.val = val;
    const float direction = gl.direction ? ch.c0 : gl.def_direction;
    const float half_len = gl.half_length ? ch.c1 : gl.def_half_length;
    float hx = half_len * (float)g.inv_csx;
    float hy = half_len * (float)g.inv_csy;
    hx = fminf(hx, gl.max_radius);
    hy = fminf(hy, gl.max_radius);
    float sd, cd;
    sincos_f32_small(direction, sd, cd);
    const float px = hx * cd, py = hy * sd;
    const double mx = (double)(fabsf(px) * 6e-7f + 1e-7f), my = (double)(fabsf(py) * 6e-7f + 1e-7f);
    bool ok = fabsf(direction) <= 64.0f;
    auto rnd = [&](double v, double m, int& out) {
        const double t = v + 0.5, fl = floor(t), f = t - fl;
        ok = ok & (f > m) & (f < 1.0 - m);
        out = (int)fl;
    };
    rnd(pg.fcx - (double)px, mx, q.ix0);
    rnd(pg.fcy - (double)py, my, q.iy0);
    rnd(pg.fcx + (double)px, mx, q.ix1);
    rnd(pg.fcy + (double)py, my, q.iy1);
    q.cx0 = pg.cx0; q.cx1 = pg.cx1; q.cy0 = pg.cy0; q.cy1 = pg.cy1;
    ambiguous = !ok;
    return q;
}

// One LANE walks one segment (integer Bresenham, glyph_kernels.cu:252-278), weight 1 per cell.
template <typename Sink>
__device__ __forceinline__ void line_walk(const LineParams& q, Sink& sink) {
    int ddx = abs(q.ix1 - q.ix0), ddy = abs(q.iy1 - q.iy0);
    int sxs = q.ix0 < q.ix1 ? 1 : -1, sys = q.iy0 < q.iy1 ? 1 : -1;
    int err = ddx - ddy, cx = q.ix0, cy = q.iy0;
    // Deliberate divergence: a segment longer than 2^23 cells (a garbage half_length or
    // direction) is dropped instead of being walked for minutes inside one lane.
    long long bound = 2ll * ((long long)ddx + ddy) + 2;
    if (ddx < 0 || ddy < 0 || bound > (1ll << 24)) return;
    int max_steps = (int)bound;
    for (int step = 0; step <= max_steps; ++step) {
        if (cx >= q.cx0 && cx < q.cx1 && cy >= q.cy0 && cy < q.cy1) sink.add(cy, cx, q.val, 1.0f);
        if (cx == q.ix1 && cy == q.iy1) break;
        int e2 = 2 * err;
        if (e2 > -ddy) { err -= ddy; cx += sxs; }
        if (e2 < ddx) { err += ddx; cy += sys; }
    }
}

// The same walk for a whole WAVE of segments at once (tile kernels): every lane visits exactly max(ddx, ddy) + 1 cells
// (a Bresenham segment advances along its major axis every step), so the loop runs to the longest segment of the wave
// with lanes masked off as they finish, and the two error updates are selects.  The per-lane form above spends more
// scalar instructions on its three data-dependent branches per step than vector instructions on the walk
// (rocprofv3: 617 SALU vs 1077 VALU wave-instructions per 64 segments in k_tile_line).
template <typename Sink>
__device__ __forceinline__ void line_walk_wave(const LineParams& q, bool valid, Sink& sink) {
    int ddx = abs(q.ix1 - q.ix0), ddy = abs(q.iy1 - q.iy0);
    const int sxs = q.ix0 < q.ix1 ? 1 : -1, sys = q.iy0 < q.iy1 ? 1 : -1;
    int err = ddx - ddy, cx = q.ix0, cy = q.iy0;
    const long long bound = 2ll * ((long long)ddx + ddy) + 2;
    if (!valid || bound > (1ll << 24)) { ddx = ddy = -1; }              // nothing to visit (garbage segments are dropped)
    int left = max(ddx, ddy) + 1;                                       // cells this lane still has to visit
    int longest = left;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) longest = max(longest, __shfl_xor(longest, off, 64));
    for (int step = 0; step < longest; ++step) {
        const bool on = left > 0;
        if (on && cx >= q.cx0 && cx < q.cx1 && cy >= q.cy0 && cy < q.cy1) sink.add(cy, cx, q.val, 1.0f);
        --left;
        const int e2 = 2 * err;
        const bool mx = e2 > -ddy, my = e2 < ddx;
        err += (mx ? -ddy : 0) + (my ? ddx : 0);
        cx += mx ? sxs : 0;
        cy += my ? sys : 0;
    }
}

// The same walk when every segment of the wave lies inside its clip rectangle AND inside the LDS window (the usual
// case: reference tiles are thousands of cells wide, segments tens): no clip test per cell, and the cell's position is
// carried as the window index li directly (li += +-1 for an x move, +-lw for a y move) instead of (cx, cy).  The tile
// kernels are bound by vector-instruction issue (a wave64 instruction holds its SIMD for four cycles); this form
// spends ~14 of them per visited cell where the general one spends ~30.  sink.add_at(li, v): window index, no checks.
template <typename Sink>
__device__ __forceinline__ void line_walk_wave_inside(const LineParams& q, bool valid, int li0, int lw, Sink& sink) {
    int ddx = abs(q.ix1 - q.ix0), ddy = abs(q.iy1 - q.iy0);
    const int stepx = q.ix0 < q.ix1 ? 1 : -1, stepy = q.iy0 < q.iy1 ? lw : -lw;
    int err = ddx - ddy, li = li0;
    if (!valid) { ddx = ddy = -1; }
    int left = max(ddx, ddy) + 1;
    int longest = left;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) longest = max(longest, __shfl_xor(longest, off, 64));
    for (int step = 0; step < longest; ++step) {
        if (left > 0) sink.add_at(li, q.val);
        --left;
        const int e2 = 2 * err;
        const bool mx = e2 > -ddy, my = e2 < ddx;
        err += (mx ? -ddy : 0) + (my ? ddx : 0);
        li += (mx ? stepx : 0) + (my ? stepy : 0);
    }
}

}  // namespace pcrhip
