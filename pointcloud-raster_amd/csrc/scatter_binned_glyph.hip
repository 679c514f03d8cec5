// scatter_binned_glyph.hip -- Gaussian / Line glyphs on LDS tiles.
//
// Points are binned by the LDS tile that contains their CENTRE cell (bin_points): index records for the
// Gaussian, 32-byte position + value + channel records for the Line (see RecordKind, engine.hpp).
// One workgroup per work item keeps an LDS window = the bin's interior plus an apron of A cells
// on every side, paints its points' footprints into it with LDS atomics (ds_add_f64 for the
// v*w and w sums of the Gaussian, ds_add_f64 + ds_add_u32 for the Line -- ds_add_f32 is ~28x
// slower on gfx950, see scatter_binned.hip), then merges the window into the HBM planes with
// float atomics, shaped as contiguous row segments (aprons of neighbouring bins overlap, so the
// merge cannot be a plain store).  A footprint cell that falls outside the window (radius larger
// than the apron the LDS can afford) goes straight to a global atomic: correct for any radius,
// fast when the apron covers it.
//
// Replaces kernel_glyph_gaussian / kernel_glyph_line (src/engine/glyph_kernels.cu:345-492);
// arithmetic follows the CPU reference (glyph_device.hpp).
#include "bin16.hpp"

#include <cstdlib>
#include <type_traits>

using namespace pcrhip;

namespace {

constexpr int kThreads = 1024;

struct GlyphTile {
    BinGeom bins;       // interior = bins.tile_w x bins.tile_h
    int apron;          // LDS apron, cells
    int lw, lh;         // LDS window = interior + 2 * apron
    int need;           // apron the glyph spec can need (>= apron when the LDS could not afford it)
    int fixed_r;        // Gaussian: the radius every point has (default sigma, no rotation, r <= 7), else 0
};

// ---- sinks ---------------------------------------------------------------------------------------
// Gaussian: the v*w sum is double in LDS (ds_add_f64, 22 cycles per wave-instruction); the WEIGHT sum is 64-bit fixed
// point with 40 fractional bits (ds_add_u64, 12 cycles).  A weight lies in [1e-6, 1] (glyph_kernels.cu:166), so the
// quantum 2^-40 = 9e-13 is below 1e-6 of the smallest weight and 2^23 full weights fit before the sum wraps -- more than a
// work item has records.  (The v*w sum cannot follow: a rim weight of 1e-6 times a small value needs a quantum that
// the largest value times the record count no longer fits.)  Measured, sigma = 1, 50 M points, WeightedAverage:
// k_tile_gauss 4.31 -> 3.65 ms.
// w * 2^40 rounded to nearest, via the 2^52 trick (exact: the product is a power-of-two scaling, the sum rounds to an integer).
__device__ __forceinline__ unsigned long long weight_fix40(float w) {
    const double d = (double)w * 1099511627776.0 + 4503599627370496.0;
    return (unsigned long long)__double_as_longlong(d) - 0x4330000000000000ull;
}
__device__ __forceinline__ double weight_unfix40(unsigned long long q) { return (double)q * (1.0 / 1099511627776.0); }

template <unsigned MASK>
struct GaussLdsSink {
    const GridDev& g;
    PlanesDev pl;
    double* t_s;
    unsigned long long* t_w;
    int x0, y0, lw, lh;            // window origin in GLOBAL cell coordinates
    __device__ __forceinline__ void add(int row, int col, float vw, float w) {
        int lx = col - x0, ly = row - y0;
        // A weight that is not in [0, 1] is a NaN (a NaN / inf per-point sigma or rotation: cos(NaN) -> NaN passes the
        // reference's `w < 1e-6f` test, glyph_kernels.cu:166): the fixed-point plane cannot hold it, so it takes the
        // float-atomic branch below and reaches the plane as the NaN the reference produces.
        if ((unsigned)lx < (unsigned)lw && (unsigned)ly < (unsigned)lh && (!(MASK & PCR_HIP_PLANE_WGT) || w <= 1.0f)) {
            int li = ly * lw + lx;
            if (MASK & PCR_HIP_PLANE_SUM) unsafeAtomicAdd(&t_s[li], (double)vw);
            if (MASK & PCR_HIP_PLANE_WGT) atomicAdd(&t_w[li], weight_fix40(w));
        } else {
            int64_t cell = (int64_t)(row - g.st_r0) * g.W + col;
            if (MASK & PCR_HIP_PLANE_SUM) atomic_add_f32(pl.sum + cell, vw);
            if (MASK & PCR_HIP_PLANE_WGT) atomic_add_f32(pl.wgt + cell, w);
        }
    }
};

// Line: weight is 1 per visited cell -> integer count.
// COVERED: the LDS apron covers the glyph's whole reach (the usual case), so a cell inside the clip rectangle is inside
// the window and the global-atomic spill branch does not exist in the walk's inner loop.
template <unsigned MASK, bool COVERED>
struct LineLdsSink {
    const GridDev& g;
    PlanesDev pl;
    double* t_s;
    unsigned* t_c;
    int x0, y0, lw, lh;
    __device__ __forceinline__ void add(int row, int col, float vw, float) {
        int lx = col - x0, ly = row - y0;
        if ((unsigned)lx < (unsigned)lw && (unsigned)ly < (unsigned)lh) {
            int li = ly * lw + lx;
            if (MASK & PCR_HIP_PLANE_SUM) unsafeAtomicAdd(&t_s[li], (double)vw);
            if (MASK & PCR_HIP_PLANE_WGT) atomicAdd(&t_c[li], 1u);
        } else if (!COVERED) {                    // (COVERED: unreachable for a finite segment; the test only guards the LDS)
            int64_t cell = (int64_t)(row - g.st_r0) * g.W + col;
            if (MASK & PCR_HIP_PLANE_SUM) atomic_add_f32(pl.sum + cell, vw);
            if (MASK & PCR_HIP_PLANE_WGT) atomic_add_f32(pl.wgt + cell, 1.0f);
        }
    }
    // a cell known to lie inside the window, by its window index (line_walk_wave_inside)
    __device__ __forceinline__ void add_at(int li, float vw) {
        if (MASK & PCR_HIP_PLANE_SUM) unsafeAtomicAdd(&t_s[li], (double)vw);
        if (MASK & PCR_HIP_PLANE_WGT) atomicAdd(&t_c[li], 1u);
    }
};

// ---- Gaussian tiles ---------------------------------------------------------------------------------
template <unsigned MASK, int FR>
__global__ void __launch_bounds__(kThreads)
k_tile_gauss(GridDev g, GlyphDev gl, GlyphTile t, PlanesDev pl, const uint2* __restrict__ records,
             const BinItem* __restrict__ items, const unsigned* __restrict__ n_items,
             const double* __restrict__ x, const double* __restrict__ y, const float* __restrict__ v) {
    extern __shared__ double lds_win[];
    if (blockIdx.x >= *n_items) return;
    const BinItem it = items[blockIdx.x];
    const int cells = t.lw * t.lh;
    double* t_s = lds_win;
    unsigned long long* t_w = reinterpret_cast<unsigned long long*>(t_s + ((MASK & 1) ? cells : 0));
    for (int i = threadIdx.x; i < cells * (((MASK & 1) ? 1 : 0) + ((MASK & 2) ? 1 : 0)); i += kThreads) lds_win[i] = 0.0;   // +0.0 = all bits zero
    __syncthreads();

    const int bx = it.bin % t.bins.bins_x, by = it.bin / t.bins.bins_x;
    GaussLdsSink<MASK> sink{g, pl, t_s, t_w, bx * t.bins.tile_w - t.apron,
                            g.st_r0 + t.bins.row0 + by * t.bins.tile_h - t.apron, t.lw, t.lh};

    // each wave takes 64 of the item's points at a time: one lane prepares one point, then all
    // 64 lanes paint the points one after the other
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint2* rec = records + it.first;
    for (unsigned j0 = wave * 64; j0 < it.count; j0 += kThreads) {
        unsigned j = j0 + lane;
        bool valid = j < it.count;
        GaussParams q{};
        if (valid) {
            // index records: this kernel is bound by LDS-atomic issue, the gather of x, y, v hides behind it
            const uint64_t i = rec[j].y;
            PointGeom pg = point_geom(g, x[i], y[i]);
            valid = pg.valid;                     // always true for a binned point; keeps q sane
            if (valid) q = gauss_params(g, gl, pg, v[i], load_chan(gl, i));
        }
        if (FR > 0) {                      // every point has radius FR: separable, fully unrolled
            if (valid) gauss_splat_fixed<FR>(q, sink);
            continue;
        }
        // small footprints (<= 11 x 11): one lane per point; larger ones: the whole wave per point
        const bool big = valid && q.r > 5;
        if (!__any(big)) {
            if (valid) gauss_splat_lane(q, sink);
        } else {
            unsigned long long todo = __ballot(valid);
            while (todo) {
                int src = __builtin_ctzll(todo);
                todo &= todo - 1;
                GaussParams u = lane_bcast(q, src);
                gauss_splat_wave(u, lane, sink);
            }
        }
    }
    __syncthreads();

    // merge: window -> planes with float atomics, consecutive lanes on consecutive cells of a row
    for (int i = threadIdx.x; i < cells; i += kThreads) {
        double s = (MASK & 1) ? t_s[i] : 0.0;
        double w = (MASK & 2) ? weight_unfix40(t_w[i]) : 0.0;
        if (s == 0.0 && w == 0.0) continue;
        int ly = i / t.lw, lx = i - ly * t.lw;
        int col = sink.x0 + lx, row = sink.y0 + ly;          // non-zero cells were clipped to the grid already
        int64_t cell = (int64_t)(row - g.st_r0) * g.W + col;
        if ((MASK & 1) && s != 0.0) atomic_add_f32(pl.sum + cell, (float)s);
        if ((MASK & 2) && w != 0.0) atomic_add_f32(pl.wgt + cell, (float)w);
    }
}

// ---- Line tiles on 16-byte records (bin16.hpp) -----------------------------------------------------------------
// Round 2 carried x, y, value and the per-point channels to the tile in 32-byte records and did the end-point
// arithmetic (an f64 sincos per segment) there.  The scatter pass of the shared front-end has x, y and the channels in
// registers anyway: it computes the segment's rounded END POINTS (glyph_kernels.cu:213-250) and stores
//     {local centre cell, value, (ix0, iy0), (ix1, iy1)}      end points as int16 offsets from the tile's origin
// -- half the record bytes both ways, and the tile kernel is left with the integer walk.  The clip rectangle (the
// centre cell's reference tile, Q4) follows from the centre cell.  A segment whose end points do not fit an int16
// offset (a garbage half_length: |offset| > 32000 cells) goes to the list and is walked by the direct form.
struct LineRecMaker {
    static constexpr bool kVectorGeometry = false;     // k_b16_scatter: grid and bin geometry in vector registers (bin16.hpp)
    static constexpr bool kCentre = false;
    static constexpr bool kOwnsX = false;
    static constexpr bool kFixup = false;
    static constexpr int kPer = 12, kBatch = 6;               // the f64 sincos of the end points needs the registers
    GlyphDev gl;
    struct Chan { float dir, hl; };
    __device__ __forceinline__ Chan load(uint64_t i) const {
        Chan c{0.f, 0.f};
        if (gl.direction) c.dir = gl.direction[i];
        if (gl.half_length) c.hl = gl.half_length[i];
        return c;
    }
    __device__ __forceinline__ bool make(const GridDev& g, const BinGeom& b, const b16::Routed16& r, const PointGeom& pg,
                                         float val, const Chan& ch, uint4& rec) const {
        const LineParams q = line_params(g, gl, pg, val, GlyphChan{ch.dir, ch.hl, 0.f});
        const int by = r.by, bx = r.bx;
        const long long ox = (long long)bx * b.tile_w, oy = (long long)g.st_r0 + b.row0 + (long long)by * b.tile_h;
        const long long a0 = q.ix0 - ox, b0 = q.iy0 - oy, a1 = q.ix1 - ox, b1 = q.iy1 - oy;
        const long long lim = 32000;
        if (a0 < -lim || a0 > lim || b0 < -lim || b0 > lim || a1 < -lim || a1 > lim || b1 < -lim || b1 > lim) return false;
        rec.y = __float_as_uint(val);
        rec.z = ((unsigned)a0 & 0xFFFFu) | ((unsigned)b0 << 16);
        rec.w = ((unsigned)a1 & 0xFFFFu) | ((unsigned)b1 << 16);
        return true;
    }
};

template <unsigned MASK, bool COVERED>
__global__ void __launch_bounds__(kThreads)
k_tile_line16(GridDev g, GlyphTile t, PlanesDev pl, const uint4* __restrict__ records,
              const BinItem* __restrict__ items, const unsigned* __restrict__ n_items) {
    extern __shared__ double lds_win[];
    if (blockIdx.x >= *n_items) return;
    const BinItem it = items[blockIdx.x];
    const int cells = t.lw * t.lh;                           // even (lw, lh even)
    double* t_s = lds_win;
    unsigned* t_c = reinterpret_cast<unsigned*>(t_s + ((MASK & 1) ? cells : 0));
    // the first records are in flight while the window is cleared
    const uint4* rec = records + it.first;
    uint4 cur = threadIdx.x < it.count ? stream_load(rec + threadIdx.x) : make_uint4(b16::kNullCell, 0u, 0u, 0u);
    for (int i = threadIdx.x; i < cells; i += kThreads) {
        if (MASK & 1) t_s[i] = 0.0;
        if (MASK & 2) t_c[i] = 0u;
    }
    __syncthreads();

    const int bx = it.bin % t.bins.bins_x, by = it.bin / t.bins.bins_x;
    const int ox = bx * t.bins.tile_w, oy = g.st_r0 + t.bins.row0 + by * t.bins.tile_h;      // the tile's origin, global cells
    LineLdsSink<MASK, COVERED> sink{g, pl, t_s, t_c, ox - t.apron, oy - t.apron, t.lw, t.lh};
    // every wave runs the same number of rounds (the walk below is wave-cooperative: shuffles inside)
    for (unsigned j0 = 0; j0 < it.count; j0 += kThreads) {
        const unsigned jn = j0 + kThreads + threadIdx.x;
        const uint4 nxt = jn < it.count ? stream_load(rec + jn) : make_uint4(b16::kNullCell, 0u, 0u, 0u);
        const bool valid = cur.x != b16::kNullCell;
        LineParams q{};
        if (valid) {
            const int ly = fast_div((int)cur.x, t.bins.tile_w), lx = (int)cur.x - ly * t.bins.tile_w;
            const int col = ox + lx, row = oy + ly;                      // the centre cell: its reference tile clips (Q4)
            const int tcx = fast_div(col, g.tw), tcy = fast_div(row, g.th);
            q.cx0 = tcx * g.tw;
            q.cx1 = min(q.cx0 + g.tw, g.W);
            q.cy0 = max(tcy * g.th, g.st_r0);
            q.cy1 = min(min(tcy * g.th + g.th, g.H), g.st_r0 + g.st_rows);
            q.val = __uint_as_float(cur.y);
            q.ix0 = ox + (int)(short)(cur.z & 0xFFFFu);
            q.iy0 = oy + (int)(short)(cur.z >> 16);
            q.ix1 = ox + (int)(short)(cur.w & 0xFFFFu);
            q.iy1 = oy + (int)(short)(cur.w >> 16);
        }
        // the whole wave's segments inside their clip rectangles and the window (bounding boxes): the short walk
        const int bx0 = min(q.ix0, q.ix1), bx1 = max(q.ix0, q.ix1), by0 = min(q.iy0, q.iy1), by1 = max(q.iy0, q.iy1);
        const bool inside = !valid || (bx0 >= max(q.cx0, sink.x0) && bx1 < min(q.cx1, sink.x0 + sink.lw) &&
                                       by0 >= max(q.cy0, sink.y0) && by1 < min(q.cy1, sink.y0 + sink.lh));
        if (__all(inside)) line_walk_wave_inside(q, valid, (q.iy0 - sink.y0) * sink.lw + (q.ix0 - sink.x0), sink.lw, sink);
        else line_walk_wave(q, valid, sink);
        cur = nxt;
    }
    __syncthreads();

    for (int i = threadIdx.x; i < cells; i += kThreads) {
        double s = (MASK & 1) ? t_s[i] : 0.0;
        unsigned c = (MASK & 2) ? t_c[i] : 0u;
        if (s == 0.0 && c == 0u) continue;
        int ly = i / t.lw, lx = i - ly * t.lw;
        int64_t cell = (int64_t)(sink.y0 + ly - g.st_r0) * g.W + (sink.x0 + lx);
        if ((MASK & 1) && s != 0.0) atomic_add_f32(pl.sum + cell, (float)s);
        if ((MASK & 2) && c) atomic_add_f32(pl.wgt + cell, (float)c);
    }
}

// ---- Line tiles on WALK-STATE records (round 4) -------------------------------------------------------------------
// profiles/r04_line_ablation.md: k_tile_line16's walk takes 1.2 ms of vector / scalar instructions WITHOUT a single LDS
// atomic in it -- ~20 instructions per 64-lane step (two error tests, four selects, a per-lane countdown) and ~185 per
// batch of 64 segments for what does not depend on the step at all (the clip rectangle from two integer divisions, the
// bounding-box test, |dx|, |dy|, the signs, a wave reduction for the longest segment).  All of the latter is a function of
// the point alone, and the scatter pass has the point in registers: it now stores the walk's STATE instead of the end points,
//     .x = first window cell (16) | cells to visit (8) | Bresenham remainder at the first cell (8)
//     .y = value      .z = 2 M (8) | 2 m (8) | major step (int8) | minor step (int8)        (steps in window cells)
// where M / m = the longer / shorter extent.  The reference's walk (glyph_kernels.cu:252-278: both error tests every step)
// visits, as its j-th cell, major = j, minor = k_j = floor((2 j m + M - 1) / 2M) -- the two strict tests make the
// minor axis advance exactly when 2 j m + M - 1 crosses a multiple of 2M (derivation in DESIGN section 3a; Count is
// bit-exact against the oracle's walk on every parity case).  So the remainder r_j = (2 j m + M - 1) mod 2M is all the
// state a step needs:   r += 2m;  if (r >= 2M) { r -= 2M; cell += major + minor } else cell += major   -- 7 instructions.
// CLIPPING (the centre cell's reference tile, Q4) keeps a CONTIGUOUS range of j (both coordinates are monotone in j): the
// scatter pass solves it for the few segments that cross their clip rectangle and stores the state at the first kept cell
// (Maker::fixup: the long way, taken after the pass's main loop by the points make() lists -- bin16.hpp, kFixup).
// Used when the LDS apron covers the glyph's reach (default half length: every kept cell lies inside the window) and the
// window's pitch fits an int8 step; the other cases keep the end-point records above.
struct LineStateMaker {
    static constexpr bool kVectorGeometry = true;     // k_b16_scatter: grid and bin geometry in vector registers (bin16.hpp)
    static constexpr bool kCentre = false;
    static constexpr bool kOwnsX = true;
    static constexpr bool kFixup = true;
    static constexpr int kPer = 12, kBatch = 6;               // the f64 sincos of the end points needs the registers
    GlyphDev gl;
    int lw, lh, apron;                                        // the tile kernel's LDS window
    struct Chan { float dir; };
    __device__ __forceinline__ Chan load(uint64_t i) const {
        Chan c{0.f};
        if (gl.direction) c.dir = gl.direction[i];
        return c;
    }
    // n / d for 0 <= n < 2^17, 1 <= d < 2^9 (extents are <= 127 cells): a float product and a correction step
    static __device__ __forceinline__ int div_small(int n, int d) {
        int q = (int)((float)n * __frcp_rn((float)d));
        const int r = n - q * d;
        q += (r >= d) - (r < 0);
        return q;
    }
    // smallest j >= 0 with k_j >= k, for m > 0:  2 j m + M - 1 >= 2 M k
    static __device__ __forceinline__ int first_j_with_k(int k, int M, int m) {
        const int num = 2 * M * k - M + 1;
        return num <= 0 ? 0 : div_small(num + 2 * m - 1, 2 * m);
    }
    // The short way (the unrolled main loop of the pass holds nothing else): single-precision sincos, and only segments that
    // lie inside their clip rectangle whole.  kLater for the others -- an end point that could round the other way with the
    // reference's correctly rounded cosine (~1e-4 of the points), a segment that crosses its clip rectangle (~2 %) -- and for
    // anything odd.  The pass is bound by vector instructions executed per point at these tile counts
    // (profiles/r04_glyph_sq_counters.md).
    __device__ __forceinline__ int make(const GridDev& g, const BinGeom& b, const b16::Routed16& r, const PointGeom& pg,
                                        float val, const Chan& ch, uint4& rec) const {
        bool ambiguous;
        const LineParams q = line_params_fast(g, gl, pg, val, GlyphChan{ch.dir, 0.f, 0.f}, ambiguous);
        const unsigned udx = (unsigned)max(q.ix0, q.ix1) - (unsigned)min(q.ix0, q.ix1), udy = (unsigned)max(q.iy0, q.iy1) - (unsigned)min(q.iy0, q.iy1);
        const int wx0 = __mul24(r.bx, b.tile_w) - apron, wy0 = g.st_r0 + b.row0 + __mul24(r.by, b.tile_h) - apron;   // window origin, global cells
        const int rx0 = max(q.cx0, wx0), rx1 = min(q.cx1, wx0 + lw), ry0 = max(q.cy0, wy0), ry1 = min(q.cy1, wy0 + lh);
        // (integer ands / ors on purpose: no short-circuit branches in the unrolled loop)
        const int inside = (int)(min(q.ix0, q.ix1) >= rx0) & (int)(max(q.ix0, q.ix1) < rx1) & (int)(min(q.iy0, q.iy1) >= ry0) & (int)(max(q.iy0, q.iy1) < ry1);
        if ((int)ambiguous | (inside ^ 1) | (int)(udx > 127u) | (int)(udy > 127u)) return b16::kLater;
        const int dx = (int)udx, dy = (int)udy;
        const bool xmajor = dx >= dy;
        const int M = xmajor ? dx : dy, m = xmajor ? dy : dx;
        const int sx = q.ix0 < q.ix1 ? 1 : -1, sy = q.iy0 < q.iy1 ? 1 : -1;
        const unsigned li = (unsigned)(__mul24(q.iy0 - wy0, lw) + (q.ix0 - wx0));
        const int stepA = xmajor ? sx : sy * lw, stepB = xmajor ? sy * lw : sx;
        rec.x = li | ((unsigned)(M + 1) << 16) | ((unsigned)max(M - 1, 0) << 24);           // all M + 1 cells, the remainder at j = 0
        rec.y = __float_as_uint(val);
        rec.z = (unsigned)max(2 * M, 1) | ((unsigned)(2 * m) << 8) | (((unsigned)stepA & 0xFFu) << 16) | ((unsigned)stepB << 24);
        return 1;
    }
    // The long way, for the points make() passes on (the pass calls it after its main loop, one such point per lane): the
    // exact end points (f64 sincos, as the reference) and the clip range.
    __device__ __forceinline__ bool fixup(const GridDev& g, const BinGeom& b, const b16::Routed16& r, const PointGeom& pg,
                                          uint64_t i, float val, uint4& rec) const {
        const Chan ch = load(i);
        const LineParams q = line_params(g, gl, pg, val, GlyphChan{ch.dir, 0.f, 0.f});
        // (unsigned differences: a garbage end point -- NaN direction -- must not overflow on its way to the list)
        const unsigned udx = (unsigned)max(q.ix0, q.ix1) - (unsigned)min(q.ix0, q.ix1), udy = (unsigned)max(q.iy0, q.iy1) - (unsigned)min(q.iy0, q.iy1);
        if (udx > 127u || udy > 127u) return false;          // (cannot happen while the apron covers the reach: the list)
        const int dx = (int)udx, dy = (int)udy;
        const bool xmajor = dx >= dy;
        const int M = xmajor ? dx : dy, m = xmajor ? dy : dx;
        const int sx = q.ix0 < q.ix1 ? 1 : -1, sy = q.iy0 < q.iy1 ? 1 : -1;
        const int by = r.by, bx = r.bx;
        const int wx0 = bx * b.tile_w - apron, wy0 = g.st_r0 + b.row0 + by * b.tile_h - apron;   // window origin, global cells
        // kept cells: the clip rectangle; it must lie inside the window wherever the segment goes (else: the list)
        const int rx0 = max(q.cx0, wx0), rx1 = min(q.cx1, wx0 + lw), ry0 = max(q.cy0, wy0), ry1 = min(q.cy1, wy0 + lh);
        const int bx0 = min(q.ix0, q.ix1), bx1 = max(q.ix0, q.ix1), by0 = min(q.iy0, q.iy1), by1 = max(q.iy0, q.iy1);
        int js = 0, je = M;
        if (!(bx0 >= rx0 && bx1 < rx1 && by0 >= ry0 && by1 < ry1)) {
            // the window must not be what cuts the segment: cells inside the clip rectangle but outside the window would be lost
            if ((bx0 < wx0 && q.cx0 < wx0) || (bx1 >= wx0 + lw && q.cx1 > wx0 + lw) ||
                (by0 < wy0 && q.cy0 < wy0) || (by1 >= wy0 + lh && q.cy1 > wy0 + lh)) return false;
            // major axis: u_j = u0 + su j in [lo, hi);  minor axis: w0 + sw k_j in [lo, hi)
            const int u0 = xmajor ? q.ix0 : q.iy0, su = xmajor ? sx : sy, ulo = xmajor ? rx0 : ry0, uhi = xmajor ? rx1 : ry1;
            const int w0 = xmajor ? q.iy0 : q.ix0, sw = xmajor ? sy : sx, wlo = xmajor ? ry0 : rx0, whi = xmajor ? ry1 : rx1;
            js = max(js, su > 0 ? ulo - u0 : u0 - (uhi - 1));
            je = min(je, su > 0 ? uhi - 1 - u0 : u0 - ulo);
            const int ka = sw > 0 ? wlo - w0 : w0 - (whi - 1), kb = sw > 0 ? whi - 1 - w0 : w0 - wlo;     // k in [ka, kb]
            if (m == 0) {
                if (ka > 0 || kb < 0) je = -1;
            } else {
                if (ka > m) je = -1;                          // (k_j never exceeds m)
                else if (ka > 0) js = max(js, first_j_with_k(ka, M, m));
                if (kb < m) je = min(je, kb < 0 ? -1 : first_j_with_k(kb + 1, M, m) - 1);
            }
        }
        rec.y = __float_as_uint(val);
        rec.w = r.lcell;
        if (je < js) { rec.x = b16::kNullCell; rec.z = 0u; return true; }       // nothing of it is kept
        int k = 0, rem = M > 0 ? M - 1 : 0;                  // the state at j = 0 ...
        if (js > 0) {                                        // ... and at the first kept cell of a clipped segment (M > 0 there)
            const int N = 2 * js * m + M - 1;
            k = div_small(N, 2 * M);
            rem = N - k * 2 * M;
        }
        const int cx = xmajor ? q.ix0 + sx * js : q.ix0 + sx * k, cy = xmajor ? q.iy0 + sy * k : q.iy0 + sy * js;
        const unsigned li = (unsigned)((cy - wy0) * lw + (cx - wx0));
        const int stepA = xmajor ? sx : sy * lw, stepB = xmajor ? sy * lw : sx;
        rec.x = li | ((unsigned)(je - js + 1) << 16) | ((unsigned)rem << 24);
        rec.z = (unsigned)(M > 0 ? 2 * M : 1) | ((unsigned)(2 * m) << 8) | (((unsigned)stepA & 0xFFu) << 16) | ((unsigned)stepB << 24);
        return true;
    }
};

// The value plane of the window as EXACT 64-bit fixed point (ds_add_u64) where the item's values allow it.
// tools/ubench_lds_patterns.hip: an LDS atomic wave-instruction costs ~(largest number of lanes on one bank pair) x 4 cycles
// as ds_add_f64 and x 2 as ds_add_u64 -- with the walk down to 7 instructions a step the kernel is bound by exactly that
// (64 segments of a wave stand on 64 unrelated cells: ~5.4 lanes on the fullest bank pair).  A float is an integer multiple
// of 2^(E - 150) (E its biased exponent), so every value whose exponent lies in [E_lo, E_lo + 22] is an exact integer
// multiple q < 2^46 of 2^(E_lo - 150), and 65 536 of them (the most an item holds) sum without overflow in 63 bits: the
// sum is EXACT, rounded once when it is merged (the f64 plane rounds every add).  E_lo is guessed from the item's first
// 1024 values (their largest exponent + 3 at the top of the range: room for values 8 x larger, and down to 2^-19 of it);
// a value outside the range -- or a NaN / inf, which must reach the plane as such -- is left out of the fixed-point sums and
// raises a flag; the item then converts its plane to doubles in place (one rounding to 53 bits) and walks the left-out
// segments again with ds_add_f64 (the count plane is not touched again).  U(0, 1) values: ~1.5 % of the items have such a
// segment; redoing the whole item instead cost the kernel 0.1 ms (an item that starts among the last ones and runs 2.6 x as
// long IS the kernel's tail).  Round 3 tried the fixed-point form on the old walk and saw nothing (the walk's own
// instructions bound it then).
// LDS layout of k_tile_line_rec, fixed so that both planes are reached with an immediate DS offset from ONE packed register:
//   [0, kLineCountBase)            static words
//   [kLineCountBase, ...)          count plane, u32 per cell          (cells <= kLineMaxCells)
//   [kLineSumBase, ...)            sum plane, 8 bytes per cell        (65 528: the largest 8-byte aligned DS offset)
constexpr int kLineCountBase = 16, kLineSumBase = 65528, kLineMaxCells = 12288;
static_assert((size_t)kLineSumBase + (size_t)kLineMaxCells * 8 <= 160 * 1024, "the Line window must fit the CU's LDS");

template <unsigned MASK>
__global__ void __launch_bounds__(kThreads)
k_tile_line_rec(GridDev g, GlyphTile t, PlanesDev pl, const uint4* __restrict__ records,
                const BinItem* __restrict__ items, const unsigned* __restrict__ n_items) {
    extern __shared__ double lds_win[];                      // (the whole segment is dynamic: no static LDS in this kernel)
    if (blockIdx.x >= *n_items) return;
    // Both planes wanted (WeightedAverage / Average): ONE 64-bit word per cell and ONE ds_add_u64 per visited cell,
    //     word = visits << 48 | sum of (q + 2^31),    q = the value as a signed multiple of 2^(e_lo - 150), |q| < 2^31
    // -- an item holds at most 65 535 records and a segment visits a cell once, so neither field can carry into the other;
    // the sum of the q is the low field minus visits * 2^31, exact.  |q| < 2^31 leaves a window of EIGHT binary exponents
    // (24 significand bits shifted by up to 7), set from the largest exponent among the item's first 1024 values: on U(0, 1)
    // values 0.4 % of the segments do not fit.  Those count in pass 0 (q = 0) and are LISTED in LDS; the item then decodes
    // its words into the count plane and a plane of doubles and walks the listed segments with ds_add_f64 (a list that
    // overflows: the whole item is scanned again for them).  The LDS pipe is the kernel's bound (29 cycles a step for
    // ds_add_u64 + ds_add_u32 on 64 unrelated cells, 12-14 for the ds_add_u64 alone: tools/ubench_lds_region.hip).
    constexpr bool PACK = MASK == 3u;
    constexpr int kWindow = PACK ? 7 : 22;                   // accepted exponents: [e_lo, e_lo + kWindow]
    const BinItem it = items[blockIdx.x];
    const int cells = t.lw * t.lh;                           // <= kLineMaxCells (checked on the host)
    char* lds = reinterpret_cast<char*>(lds_win);
    int* s_ehi = reinterpret_cast<int*>(lds);
    int* s_redo = s_ehi + 1;                                 // 1: some segment was left out of the fixed-point sums; 2: and the list is full
    unsigned* s_nlist = reinterpret_cast<unsigned*>(s_ehi + 2);
    unsigned* t_c = reinterpret_cast<unsigned*>(lds + kLineCountBase);
    unsigned long long* t_q = reinterpret_cast<unsigned long long*>(lds + kLineSumBase);
    double* t_s = reinterpret_cast<double*>(lds + kLineSumBase);
    // the left-out segments of the item (record numbers): between the count plane and the sum plane
    unsigned* list = t_c + cells;
    const unsigned list_cap = (unsigned)((kLineSumBase - kLineCountBase) / 4 - cells);
    const uint4* rec = records + it.first;
    const uint4 first = threadIdx.x < it.count ? stream_load(rec + threadIdx.x) : make_uint4(b16::kNullCell, 0u, 0u, 0u);
    if (threadIdx.x == 0) { *s_ehi = 0; *s_redo = 0; *s_nlist = 0u; }
    for (int i = threadIdx.x; i < cells; i += kThreads) {
        if (MASK & 1) t_q[i] = 0ull;
        if ((MASK & 2) && !PACK) t_c[i] = 0u;
    }
    __syncthreads();
    int e_lo = 0;
    if (MASK & 1) {
        int e = 0;
        if (first.x != b16::kNullCell) {
            e = (int)((first.y >> 23) & 0xFFu);
            if (e == 255) e = 0;                             // NaN / inf: found again (and listed) by the walk
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) e = max(e, __shfl_xor(e, off, 64));
        if ((threadIdx.x & 63) == 0 && e > 0) atomicMax(s_ehi, e);
        __syncthreads();
        // unpacked: room for values 8 x larger than the first 1024, and down to 2^-19 of them; packed: the window is too
        // short for head room -- a later, larger value is listed
        e_lo = PACK ? *s_ehi - kWindow : *s_ehi + 3 - kWindow;
    }
    auto fits_window = [&](unsigned bits) {
        const int e = (int)((bits >> 23) & 0xFFu);
        return (bits & 0x7FFFFFFFu) == 0u || (e != 255 && e >= max(e_lo, 1) && e <= e_lo + kWindow);
    };
    // The walk's whole state in ONE register: st = (byte offset of the cell in the count plane) << 16 | remainder.
    //   minor step taken  <=>  remainder >= 2M - 2m;   st += taken ? d1 : d0
    //   d0 = (4 major) << 16 | 2m        d1 = (4 (major + minor)) << 16 + 2m - 2M        (plain 32-bit adds: no field borrows)
    // Both plane addresses are shifts of st (the remainder stays below 2^15, so bit 15 -- bit 0 of st >> 15 -- is clear).
    // Up to the shortest segment of the wave no lane needs masking; the steps there are unrolled by four.
    // pass 0: visits + fixed-point sums of the values that fit; pass 1 (only when some did not): those, as doubles
    auto walk_wave = [&](auto pass_c, const uint4 cur, bool valid, unsigned recno) {
        constexpr int PASS = decltype(pass_c)::value;
        bool fits = true;
        if (MASK & 1) fits = fits_window(cur.y);
        if (PASS == 1) valid = valid && !fits;               // the second pass walks only what the first left out
        const unsigned n = valid ? (cur.x >> 16) & 0xFFu : 0u;
        const unsigned M2 = cur.z & 0xFFu, m2 = (cur.z >> 8) & 0xFFu;
        const int sA4 = 4 * (int)(signed char)((cur.z >> 16) & 0xFFu), sB4 = 4 * (int)(signed char)(cur.z >> 24);
        unsigned st = ((cur.x & 0xFFFFu) << 18) | (cur.x >> 24);
        const unsigned thr = M2 - m2;
        const unsigned d0 = ((unsigned)sA4 << 16) + m2, d1 = ((unsigned)(sA4 + sB4) << 16) + m2 - M2;
        const double dv = (double)__uint_as_float(cur.y);
        unsigned long long q = 0ull;
        if ((MASK & 1) && PASS == 0) {
            const int e = (int)((cur.y >> 23) & 0xFFu);
            const bool zero = (cur.y & 0x7FFFFFFFu) == 0u;
            if (n > 0 && !fits) {
                const unsigned k = atomicAdd(s_nlist, 1u);
                if (k < list_cap) list[k] = recno;
                atomicMax(s_redo, k < list_cap ? 1 : 2);
            }
            const long long mag = zero || !fits ? 0ll : (long long)((cur.y & 0x7FFFFFu) | 0x800000u) << (e - e_lo);
            q = (unsigned long long)((cur.y >> 31) ? -mag : mag);
            if (PACK) q += (1ull << 48) + (1ull << 31);
        }
        auto step = [&]() {
            if (MASK & 1) {
                char* ps = lds + kLineSumBase + (st >> 15);
                if (PASS == 0) atomicAdd(reinterpret_cast<unsigned long long*>(ps), q);
                else unsafeAtomicAdd(reinterpret_cast<double*>(ps), dv);
            }
            if ((MASK & 2) && !PACK && PASS == 0) atomicAdd(reinterpret_cast<unsigned*>(lds + kLineCountBase + (st >> 16)), 1u);
        };
        auto advance = [&]() { st += (st & 0xFFFFu) >= thr ? d1 : d0; };
        // the shortest and the longest walk of the wave (null records: none / 0)
        unsigned nmin = valid ? n : 0xFFFFu, nmax = n;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            nmin = min(nmin, (unsigned)__shfl_xor((int)nmin, off, 64));
            nmax = max(nmax, (unsigned)__shfl_xor((int)nmax, off, 64));
        }
        nmin = __builtin_amdgcn_readfirstlane(nmin);
        nmax = __builtin_amdgcn_readfirstlane(nmax);
        if (nmin > nmax) nmin = 0;                        // (a wave of null records)
        unsigned j = 0;
        if (valid) {
            for (; j + 4 <= nmin; j += 4) {
                step(); advance(); step(); advance(); step(); advance(); step(); advance();
            }
            for (; j < nmin; ++j) { step(); advance(); }
        }
        j = nmin;
        for (; j < nmax; ++j) {
            if (j < n) step();
            advance();
        }
    };
    auto walk = [&](auto pass_c) {
        uint4 cur = first;
        for (unsigned j0 = 0; j0 < it.count; j0 += kThreads) {
            const unsigned jn = j0 + kThreads + threadIdx.x;
            const uint4 nxt = jn < it.count ? stream_load(rec + jn) : make_uint4(b16::kNullCell, 0u, 0u, 0u);
            walk_wave(pass_c, cur, cur.x != b16::kNullCell, j0 + threadIdx.x);
            cur = nxt;
        }
    };
    walk(std::integral_constant<int, 0>{});
    __syncthreads();
    const double scale = ldexp(1.0, e_lo - 150);
    auto unpack_sum = [&](unsigned long long w) {             // the sum of the q of a packed word
        return (long long)(w & 0xFFFFFFFFFFFFull) - ((long long)(w >> 48) << 31);
    };
    const int redo = (MASK & 1) ? *s_redo : 0;
    if (redo) {
        for (int i = threadIdx.x; i < cells; i += kThreads) {
            const unsigned long long w = t_q[i];
            if (PACK) t_c[i] = (unsigned)(w >> 48);
            t_s[i] = (double)(PACK ? unpack_sum(w) : (long long)w) * scale;
        }
        __syncthreads();
        if (redo == 1) {                                      // the listed segments, one per lane (their records come back from the L2)
            const unsigned nl = *s_nlist;
            for (unsigned k0 = 0; k0 < nl; k0 += kThreads) {
                const unsigned k = k0 + threadIdx.x;
                uint4 cur = make_uint4(b16::kNullCell, 0u, 0u, 0u);
                if (k < nl) cur = rec[list[k]];
                walk_wave(std::integral_constant<int, 1>{}, cur, cur.x != b16::kNullCell, 0u);
            }
        } else {
            walk(std::integral_constant<int, 1>{});
        }
        __syncthreads();
    }
    const int bx = it.bin % t.bins.bins_x, by = it.bin / t.bins.bins_x;
    const int x0 = bx * t.bins.tile_w - t.apron, y0 = t.bins.row0 + by * t.bins.tile_h - t.apron;      // window origin; rows relative to the state window
    for (int i = threadIdx.x; i < cells; i += kThreads) {
        double s = 0.0;
        unsigned c = 0u;
        if (redo) {
            if (MASK & 1) s = t_s[i];
            if (MASK & 2) c = t_c[i];
        } else if (PACK) {
            const unsigned long long w = t_q[i];
            c = (unsigned)(w >> 48);
            s = (double)unpack_sum(w) * scale;
        } else {
            if (MASK & 1) s = (double)(long long)t_q[i] * scale;
            if (MASK & 2) c = t_c[i];
        }
        if (s == 0.0 && c == 0u) continue;
        int ly = i / t.lw, lx = i - ly * t.lw;
        int64_t cell = (int64_t)(y0 + ly) * g.W + (x0 + lx);     // non-zero cells were clipped to the grid by the scatter pass
        if ((MASK & 1) && s != 0.0) atomic_add_f32(pl.sum + cell, (float)s);
        if ((MASK & 2) && c) atomic_add_f32(pl.wgt + cell, (float)c);
    }
}

// the listed segments: one lane walks one segment straight into the planes
template <unsigned MASK>
struct LineDirectSink {
    const GridDev& g;
    PlanesDev pl;
    __device__ __forceinline__ void add(int row, int col, float vw, float) {
        const int64_t cell = (int64_t)(row - g.st_r0) * g.W + col;
        if (MASK & 1) atomic_add_f32(pl.sum + cell, vw);
        if (MASK & 2) atomic_add_f32(pl.wgt + cell, 1.0f);
    }
};

template <unsigned MASK>
__global__ void __launch_bounds__(256)
k_line_list(GridDev g, GlyphDev gl, PlanesDev pl, const unsigned* __restrict__ list, const unsigned* __restrict__ count,
            const double* __restrict__ x, const double* __restrict__ y, const float* __restrict__ v) {
    const unsigned n = *count;
    LineDirectSink<MASK> sink{g, pl};
    for (unsigned j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const uint64_t i = list[j];
        PointGeom pg = point_geom(g, x[i], y[i]);
        if (!pg.valid) continue;
        const LineParams q = line_params(g, gl, pg, v[i], load_chan(gl, i));
        line_walk(q, sink);
    }
}

// ---- host: tile geometry from the glyph spec ---------------------------------------------------------
int apron_needed(const GridDev& g, const GlyphDev& gl) {
    double cap = std::min<double>(std::max(gl.max_radius, 0.0f), 4096.0);
    if (gl.type == PCR_HIP_GLYPH_GAUSSIAN) {
        if (gl.sigma_x || gl.sigma_y) return (int)std::ceil(cap);
        float sx = gl.def_sigma_x * (float)g.inv_csx, sy = gl.def_sigma_y * (float)g.inv_csy;
        float R = std::min(3.0f * std::max(sx, sy), gl.max_radius);
        if (!(R >= 0.0f)) return 0;
        return (int)std::ceil(std::min<double>(R, 4096.0));
    }
    // Line: reach of the rounded end points
    if (gl.half_length) return (int)std::ceil(cap) + 1;
    double hx = std::min<double>(std::fabs(gl.def_half_length * (float)g.inv_csx), 4096.0);
    double hy = std::min<double>(std::fabs(gl.def_half_length * (float)g.inv_csy), 4096.0);
    if (gl.def_half_length * (float)g.inv_csx > gl.max_radius) hx = cap;       // std::min(h, cap) caps positive h only
    if (gl.def_half_length * (float)g.inv_csy > gl.max_radius) hy = cap;
    return (int)std::ceil(std::max(hx, hy)) + 1;
}

int cell_bytes(int glyph_type, unsigned mask) {
    if (glyph_type == PCR_HIP_GLYPH_GAUSSIAN) return ((mask & 1) ? 8 : 0) + ((mask & 2) ? 8 : 0);
    return ((mask & 1) ? 8 : 0) + ((mask & 2) ? 4 : 0);
}

// Tile shape from the glyph's reach: the interior S x S is what the LDS leaves after an apron of `need`
// cells on every side.  A grid with more such tiles than the binning passes take is swept in row bands
// (band_rows rows each); only when even kMaxBands bands are not enough is S grown at the apron's expense.
bool glyph_tile(const pcr_hip_engine* e, const GlyphDev& gl, unsigned mask, GlyphTile* out, int* band_rows) {
    const GridDev& g = e->gd;
    const int limit = 150 * 1024 / std::max(cell_bytes(gl.type, mask), 4);    // LDS cells per workgroup
    const int side = ((int)std::floor(std::sqrt((double)limit))) & ~1;
    const int need = apron_needed(g, gl);
    int S = std::min(128, (side - 2 * need) & ~7);
    if (gl.type == PCR_HIP_GLYPH_LINE && !gl.half_length) {
        // the walk-state tile kernel (k_tile_line_rec) wants the whole window within kLineMaxCells cells whatever the planes:
        // one tile shape for Sum, Count and WeightedAverage
        const int side_rec = ((int)std::floor(std::sqrt((double)kLineMaxCells))) & ~1;           // 110
        if (side_rec - 2 * need >= 32) S = std::min(S, (side_rec - 2 * need) & ~7);
    }
    if (S < 32) S = 32;
    auto bands_for = [&](int s) {
        const int br = band_rows_for(g, s, s, e->max_bins);
        return br > 0 ? (g.st_rows + br - 1) / br : 1 << 30;
    };
    while (bands_for(S) > kMaxBands && S < side - 8) S += 8;
    if (bands_for(S) > kMaxBands) return false;
    GlyphTile t;
    t.need = need;
    t.fixed_r = 0;
    if (gl.type == PCR_HIP_GLYPH_GAUSSIAN && !gl.sigma_x && !gl.sigma_y && !gl.rotation && gl.def_rotation == 0.0f) {
        // the r of gauss_params (glyph_device.hpp) for the default sigmas
        const float sx = gl.def_sigma_x * (float)g.inv_csx, sy = gl.def_sigma_y * (float)g.inv_csy;
        const float R = std::fmin(3.0f * std::fmax(sx, sy), gl.max_radius);
        const int r = std::min((int)std::ceil(R), 1 << 20);
        if (R == R && r >= 1 && r <= 7) t.fixed_r = r;      // r = 6, 7: sigmas too small for the moment path's cut-off condition
    }
    t.apron = std::max(0, std::min(need, (side - S) / 2));
    t.lw = S + 2 * t.apron;
    t.lh = S + 2 * t.apron;
    t.bins.tile_w = S;
    t.bins.tile_h = S;
    t.bins.bins_x = (g.W + S - 1) / S;
    *band_rows = band_rows_for(g, S, S, e->max_bins);
    t.bins.row0 = 0;
    t.bins.rows = g.st_rows;
    t.bins.sup_shift = 0;
    t.bins.bins_y = (g.st_rows + S - 1) / S;
    t.bins.nbins = t.bins.bins_x * t.bins.bins_y;
    t.bins.chunk = t.bins.nbins <= 2048 ? 16384 : 8192;
    *out = t;
    return true;
}

template <typename K>
void launch_gauss(K kernel, pcr_hip_engine* e, const GridDev& gd, const GlyphDev& gl, const GlyphTile& t, const PlanesDev& pl,
                  const BinBuffers& bb, size_t lds, const double* x, const double* y, const float* v) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3(bb.max_items), dim3(kThreads), lds, e->stream, gd, gl, t, pl, bb.records,
                       bb.items, bb.n_items, x, y, v);
}

}  // namespace

namespace pcrhip {

bool binned_glyph_supported(const pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask) {
    if (mask == 0 || (mask & ~3u)) return false;
    if (gl.type != PCR_HIP_GLYPH_GAUSSIAN && gl.type != PCR_HIP_GLYPH_LINE) return false;
    GlyphTile t;
    int band_rows = 0;
    if (!glyph_tile(e, gl, mask, &t, &band_rows)) return false;
    // every bin's window is swept once per scatter: not worth it for a handful of points
    uint64_t cells = (uint64_t)e->gd.W * e->gd.st_rows;
    if (e->forced_path != 2 && e->stats.points_in * 64 < cells) return false;
    return e->stats.points_in < (1ull << 32) - (1ull << 20);
}

int binned_glyph(pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask, const PlanesDev& pl,
                 const double* x, const double* y, const float* v, uint64_t n) {
    GlyphTile t;
    int band_rows = 0;
    if (!glyph_tile(e, gl, mask, &t, &band_rows)) return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_glyph: grid cannot be binned");
    // work item size: ~4 M cell updates per workgroup for the Gaussian, 64 K segments for the Line
    unsigned item_points = 65535;                             // (k_tile_line_rec counts a cell's visits in 16 bits)
    if (gl.type == PCR_HIP_GLYPH_GAUSSIAN) {
        double fp = (2.0 * t.need + 1.0) * (2.0 * t.need + 1.0);
        item_points = (unsigned)std::min(65536.0, std::max(1024.0, 4.0e6 / fp));
    }
    const size_t lds = (size_t)t.lw * t.lh * cell_bytes(gl.type, mask);
    const int S = t.bins.tile_h;
    int total_bins = 0;
    // Gaussian on a window with more tiles than one pass counts: the 8-byte index records go through the same
    // two-level sort as the Point glyph (one sweep instead of one per row band)
    if (gl.type == PCR_HIP_GLYPH_GAUSSIAN && band_rows < e->gd.st_rows) {
        const int shift = two_level_shift(e, t.bins.nbins);
        if (shift > 0) {
            t.bins.sup_shift = shift;
            BinBuffers bb{};
            int rc = bin_points_two_level(e, t.bins, x, y, nullptr, n, true, item_points, &bb);
            if (rc) return rc;
            {
                ScopedKernelTimer tm(e, "k_tile_gauss");
#define PCR_GAUSS(M, FR) launch_gauss(&k_tile_gauss<M, FR>, e, e->gd, gl, t, pl, bb, lds, x, y, v)
#define PCR_GAUSS_R(M)                                                                          \
                switch (t.fixed_r) {                                                            \
                    case 1: PCR_GAUSS(M, 1); break;                                             \
                    case 2: PCR_GAUSS(M, 2); break;                                             \
                    case 3: PCR_GAUSS(M, 3); break;                                             \
                    case 4: PCR_GAUSS(M, 4); break;                                             \
                    case 5: PCR_GAUSS(M, 5); break;                                             \
                    case 6: PCR_GAUSS(M, 6); break;                                             \
                    case 7: PCR_GAUSS(M, 7); break;                                             \
                    default: PCR_GAUSS(M, 0); break;                                            \
                }
                if (mask == 1) { PCR_GAUSS_R(1) } else if (mask == 2) { PCR_GAUSS_R(2) } else { PCR_GAUSS_R(3) }
#undef PCR_GAUSS_R
#undef PCR_GAUSS
            }
            PCR_HIP_TRY(hipGetLastError());
            e->stats.path = 1;
            e->stats.lds_tile_w = t.bins.tile_w;
            e->stats.lds_tile_h = t.bins.tile_h;
            e->stats.lds_apron = t.apron;
            e->stats.num_bins = t.bins.nbins;
            return PCR_HIP_OK;
        }
    }
    if (gl.type == PCR_HIP_GLYPH_LINE) {
        // 16-byte end-point records through the shared front-end (bin16.hpp); bands as below
        const int band16 = band_rows_for(e->gd, S, S, b16::max_bins(e));
        if (band16 <= 0) return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_glyph: grid cannot be binned");
        const int max_bins = t.bins.bins_x * ((std::min(band16, e->gd.st_rows) + S - 1) / S);
        const b16::Layout L = b16::layout(0, max_bins, n, item_points);
        int rc = ensure_scratch(e, L.end);
        if (rc) return rc;
        PCR_HIP_TRY(hipMemsetAsync(e->d_scratch + L.o_fbc, 0, 4, e->stream));
        const bool covered = t.apron >= t.need && !gl.half_length;
        // walk-state records: every kept cell lies inside the window, the window's pitch fits an int8 step
        const bool state_records = covered && t.lw <= 127 && t.lw * t.lh <= kLineMaxCells;
        for (int row0 = 0; row0 < e->gd.st_rows; row0 += band16) {
            const int rows = std::min(band16, e->gd.st_rows - row0);
            GridDev gd = e->gd;
            gd.own_r0 = std::max(e->gd.own_r0, e->gd.st_r0 + row0);
            gd.own_r1 = std::min(e->gd.own_r1, e->gd.st_r0 + row0 + rows);
            if (gd.own_r0 >= gd.own_r1) continue;
            t.bins.row0 = row0;
            t.bins.rows = rows;
            t.bins.bins_y = (rows + S - 1) / S;
            t.bins.nbins = t.bins.bins_x * t.bins.bins_y;
            t.bins.chunk = state_records ? b16::chunk_of<LineStateMaker>() : b16::chunk_of<LineRecMaker>();
            total_bins += t.bins.nbins;
            b16::Buffers bb{};
            if (state_records) rc = b16::bin(e, gd, t.bins, LineStateMaker{gl, t.lw, t.lh, t.apron}, x, y, v, n, item_points, L, &bb);
            else rc = b16::bin(e, gd, t.bins, LineRecMaker{gl}, x, y, v, n, item_points, L, &bb);
            if (rc) return rc;
            ScopedKernelTimer tm(e, "k_tile_line");
            auto go = [&](auto kernel) {
                const size_t bytes = state_records ? (size_t)kLineSumBase + ((mask & 1) ? (size_t)t.lw * t.lh * 8 : 0) : lds;
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
                hipLaunchKernelGGL(kernel, dim3(bb.max_items), dim3(kThreads), bytes, e->stream, e->gd, t, pl, bb.records, bb.items, bb.n_items);
            };
#define PCR_LINE16(M) if (state_records) go(&k_tile_line_rec<M>); else if (covered) go(&k_tile_line16<M, true>); else go(&k_tile_line16<M, false>);
            switch (mask) {
                case 1: PCR_LINE16(1) break;
                case 2: PCR_LINE16(2) break;
                default: PCR_LINE16(3) break;
            }
#undef PCR_LINE16
        }
        {
            ScopedKernelTimer tm(e, "k_line_list");                // normally an empty list
            unsigned* d_fbl = reinterpret_cast<unsigned*>(e->d_scratch + L.o_fbl);
            unsigned* d_fbc = reinterpret_cast<unsigned*>(e->d_scratch + L.o_fbc);
            if (mask == 1) hipLaunchKernelGGL(k_line_list<1>, dim3(64), dim3(256), 0, e->stream, e->gd, gl, pl, d_fbl, d_fbc, x, y, v);
            else if (mask == 2) hipLaunchKernelGGL(k_line_list<2>, dim3(64), dim3(256), 0, e->stream, e->gd, gl, pl, d_fbl, d_fbc, x, y, v);
            else hipLaunchKernelGGL(k_line_list<3>, dim3(64), dim3(256), 0, e->stream, e->gd, gl, pl, d_fbl, d_fbc, x, y, v);
        }
        PCR_HIP_TRY(hipGetLastError());
        e->stats.path = 1;
        e->stats.lds_tile_w = t.bins.tile_w;
        e->stats.lds_tile_h = t.bins.tile_h;
        e->stats.lds_apron = t.apron;
        e->stats.num_bins = total_bins;
        return PCR_HIP_OK;
    }
    // row bands: a band bins the points whose CENTRE row it holds; footprints reach into neighbouring
    // bands through the apron / global-atomic spill exactly as they reach into neighbouring tiles
    for (int row0 = 0; row0 < e->gd.st_rows; row0 += band_rows) {
        const int rows = std::min(band_rows, e->gd.st_rows - row0);
        GridDev gd = e->gd;
        gd.own_r0 = std::max(e->gd.own_r0, e->gd.st_r0 + row0);
        gd.own_r1 = std::min(e->gd.own_r1, e->gd.st_r0 + row0 + rows);
        if (gd.own_r0 >= gd.own_r1) continue;
        t.bins.row0 = row0;
        t.bins.rows = rows;
        t.bins.bins_y = (rows + S - 1) / S;
        t.bins.nbins = t.bins.bins_x * t.bins.bins_y;
        t.bins.chunk = t.bins.nbins <= 2048 ? 16384 : 8192;
        total_bins += t.bins.nbins;
        BinBuffers bb{};
        int rc = bin_points(e, gd, t.bins, x, y, v, n, RecordKind::Index, &gl, item_points, &bb);
        if (rc) return rc;
        // the tile kernel re-derives every point's geometry from the engine's grid: the band only selected them
        ScopedKernelTimer tm(e, "k_tile_gauss");
#define PCR_GAUSS(M, FR) launch_gauss(&k_tile_gauss<M, FR>, e, e->gd, gl, t, pl, bb, lds, x, y, v)
#define PCR_GAUSS_R(M)                                                                          \
        switch (t.fixed_r) {                                                                    \
            case 1: PCR_GAUSS(M, 1); break;                                                     \
            case 2: PCR_GAUSS(M, 2); break;                                                     \
            case 3: PCR_GAUSS(M, 3); break;                                                     \
            case 4: PCR_GAUSS(M, 4); break;                                                     \
            case 5: PCR_GAUSS(M, 5); break;                                                     \
            case 6: PCR_GAUSS(M, 6); break;                                                     \
            case 7: PCR_GAUSS(M, 7); break;                                                     \
            default: PCR_GAUSS(M, 0); break;                                                    \
        }
        if (mask == 1) { PCR_GAUSS_R(1) } else if (mask == 2) { PCR_GAUSS_R(2) } else { PCR_GAUSS_R(3) }
#undef PCR_GAUSS_R
#undef PCR_GAUSS
    }
    PCR_HIP_TRY(hipGetLastError());
    e->stats.path = 1;
    e->stats.lds_tile_w = t.bins.tile_w;
    e->stats.lds_tile_h = t.bins.tile_h;
    e->stats.lds_apron = t.apron;
    e->stats.num_bins = total_bins;
    return PCR_HIP_OK;
}

}  // namespace pcrhip
