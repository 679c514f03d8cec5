// scatter_cells.hip -- glyph tiles that never gather and (Gaussian) hardly touch an LDS atomic.
//
// Round 2's glyph tiles were bound by two things the counters named (profiles/r02_gauss1_rocprof.md,
// r02_moments_split.md): the Gaussian tile fetched x, y, v BY INDEX (19 GB of line fetches for 1 GB of input), and
// every footprint cell of every point was its own LDS atomic at a random bank (~23 cycles per wave-instruction, 98 of
// them per point at sigma = 1).  This file replaces both for default-sigma, unrotated Gaussians of radius <= 3:
//
//  * ONE binning front-end on 16-byte VALUE records ("b16"): a counting pass that reads x, y only and writes no keys,
//    a scan, and a scatter pass that routes again from the x, y it has to read anyway (it needs the sub-cell offsets)
//    and stores finished records {local cell, value, sub-cell x, sub-cell y} straight from registers.  Small tiles
//    mean a (workgroup, tile) run is one or two records, so nothing is staged in LDS; instead every tile's record
//    range is split into EIGHT sub-ranges, one per virtual XCD (blockIdx % 8 in both passes: workgroups are dealt to
//    the XCDs round-robin), so that a 128-byte line of records is only ever written through ONE XCD's L2 and leaves
//    it whole.  (Lines shared between XCDs left as partial writes: 1.25x write amplification on round 2's 32-byte
//    records.)  The mapping is a speed matter only: any placement gives the same records.
//
//  * the tile kernel sorts its records by CELL inside LDS (counting sort; the records themselves are staged, 12 bytes
//    each, so no second trip to memory), then a LANE OWNS A CELL: the (2R+1)^2 footprints of the cell's points are
//    summed in REGISTERS (all points of a cell share the footprint's position), the 64 lanes of a wave own 64
//    consecutive cells of a row, so the footprints of neighbouring lanes overlap column-wise and are combined with
//    2R wave shifts (v_add_f32_dpp wave_shr:1) -- after which a lane holds ONE finished sum per footprint row and
//    plane, and issues it as a conflict-free ds_add_f64 (consecutive lanes, consecutive cells).  Per cell that is
//    2(2R+1) LDS atomics instead of 2(2R+1)^2 per POINT: ~20x fewer at three points per cell.
//
// Weights: w(dx, dy) = wx[dx] * wy[dy] with wx[d] = exp(-(d - u)^2 / 2 sx^2) evaluated as G_d * q^d * E0
// (G_d = exp(-d^2 / 2 sx^2) from the host, q = exp(u / sx^2), E0 = exp(-u^2 / 2 sx^2)): three exponentials per axis and
// point.  The product differs from the reference's single expf (glyph_kernels.cu:157-166) by a few ulp -- the same
// class as the reference's own rounding of (rdx/sx)^2 -- and the 1e-6 cut-off is applied to it; tested to the
// rtol 1e-4 every Gaussian path is tested to.  Points the scheme cannot represent (non-finite value, centre cell not
// the routed cell at a grid edge) go to a list and are painted by the wave-per-point direct form afterwards.
//
// Replaces kernel_glyph_gaussian (src/engine/glyph_kernels.cu:345-422) for these glyphs; arithmetic follows
// accumulate_glyph_gaussian_cpu (:79-183).
#include "bin16.hpp"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

using namespace pcrhip;
using namespace pcrhip::b16;

namespace {

// ---- pass B: records straight from registers -------------------------------------------------------------------
// Gaussian cell records {local cell, value, sub-cell x, sub-cell y}: sub = (float)(fc - floor(fc)), the reference's
// f32 sub-cell offset (glyph_kernels.cu:116-117).
struct GaussCellMaker {
    static constexpr bool kVectorGeometry = false;     // k_b16_scatter: grid and bin geometry in vector registers (bin16.hpp)
    static constexpr bool kCentre = true;
    static constexpr bool kOwnsX = false;
    static constexpr bool kFixup = false;
    static constexpr int kPer = 16, kBatch = 16;
    struct Chan {};
    __device__ __forceinline__ Chan load(uint64_t) const { return Chan{}; }
    __device__ __forceinline__ bool make(const GridDev&, const BinGeom&, const Routed16&, const PointGeom& pg, float val,
                                         const Chan&, uint4& rec) const {
        if (!finite_f(val)) return false;
        rec.y = __float_as_uint(val);
        rec.z = __float_as_uint((float)(pg.fcx - floor(pg.fcx)));
        rec.w = __float_as_uint((float)(pg.fcy - floor(pg.fcy)));
        return true;
    }
};

// ---- the listed points: wave-per-point direct splat ------------------------------------------------------------
template <unsigned MASK>
struct DirectSink {
    const GridDev& g;
    PlanesDev pl;
    __device__ __forceinline__ void add(int row, int col, float vw, float w) {
        const int64_t cell = (int64_t)(row - g.st_r0) * g.W + col;
        if (MASK & 1) atomic_add_f32(pl.sum + cell, vw);
        if (MASK & 2) atomic_add_f32(pl.wgt + cell, w);
    }
};

template <unsigned MASK>
__global__ void __launch_bounds__(256)
k_cell_gauss_list(GridDev g, GlyphDev gl, PlanesDev pl, const unsigned* __restrict__ list, const unsigned* __restrict__ count,
                  const double* __restrict__ x, const double* __restrict__ y, const float* __restrict__ v) {
    const unsigned n = *count;
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    DirectSink<MASK> sink{g, pl};
    for (unsigned base = wave * 64; base < n; base += nwaves * 64) {
        const unsigned j = base + lane;
        bool valid = j < n;
        GaussParams q{};
        if (valid) {
            const uint64_t i = list[j];
            PointGeom pg = point_geom(g, x[i], y[i]);
            valid = pg.valid;
            if (valid) q = gauss_params(g, gl, pg, v[i], load_chan(gl, i));
        }
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            GaussParams u = lane_bcast(q, src);
            gauss_splat_wave(u, lane, sink);
        }
    }
}

// ---- Gaussian cell tiles ---------------------------------------------------------------------------------------
struct CellGauss {
    float inv_s2x, inv_s2y;        // 1 / sx^2, 1 / sy^2 (cells^-2; sx, sy as the reference forms them in f32)
    float gx[4], gy[4];            // exp(-d^2 / 2 s^2), d = 0..3
    int cap;                       // records per work item
};

typedef float pcr_f2 __attribute__((ext_vector_type(2)));

// lane i <- lane i - 1 (lane 0 <- 0): the compiler folds it into v_add_f32_dpp wave_shr:1
__device__ __forceinline__ float wave_shr1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}

// wx[j] = exp(-((j - R) - u)^2 / 2 s^2), j = 0..2R, as G_d * q^d * E0 (see the file comment)
template <int R>
__device__ __forceinline__ void axis_weights(float u, float inv_s2, const float (&G)[4], float (&w)[2 * R + 1], float scale) {
    const float t = u * inv_s2;
    const float e0 = scale * __expf(-0.5f * u * t);            // scale: 1, or 0 for the null point of an idle lane
    const float q = __expf(t), qi = __expf(-t);
    w[R] = e0;
    float qp = q, qm = qi;
#pragma unroll
    for (int d = 1; d <= R; ++d) {
        const float ge = e0 * G[d];
        w[R + d] = ge * qp;
        w[R - d] = ge * qm;
        qp *= q;
        qm *= qi;
    }
}

// CUT: which cells of the footprint can fall under the reference's 1e-6 cut-off (decided on the host from the sigmas):
// 0 none, 1 the four corners only, 2 any.
// THREADS: R = 3 keeps 98 accumulator registers per lane (two waves per SIMD): two 256-thread workgroups per CU on 20-row
// tiles (one's record loads, sort and merge behind the other's arithmetic) or one of 512 on 40-row tiles; smaller R:
// 512 / 1024 threads.
template <int R, unsigned MASK, int CUT, int THREADS>
__global__ void __launch_bounds__(THREADS, R == 3 ? 2 : 4)     // R = 3: 49 accumulator pairs per lane, two waves per SIMD
k_cell_gauss(GridDev g, BinGeom b, CellGauss P, PlanesDev pl, const uint4* __restrict__ records,
             const BinItem* __restrict__ items, const unsigned* __restrict__ n_items) {
    constexpr int D = 2 * R + 1;
    constexpr int kTileThreads = THREADS, kRecPerThread = THREADS == 1024 ? 8 : 16;  // an item holds <= that many x THREADS records
    // (declared as double: the dynamic segment follows the static words below and must be 8-byte aligned for ds_add_f64)
    extern __shared__ double lds_raw[];
    if (blockIdx.x >= *n_items) return;
    const BinItem it = items[blockIdx.x];
    const int cells = b.tile_w * b.tile_h;                        // tile_w = 64 - 2R
    const int wrows = b.tile_h + 2 * R, wcells = wrows * 64;      // window: 64 columns = tile_w + 2R
    // layout: win_s f64 [wcells] | win_w f64 [wcells] | off u32 [cells + 1] | sv, ssx, ssy f32 [cap]
    double* win_s = lds_raw;
    double* win_w = win_s + ((MASK & 1) ? wcells : 0);
    unsigned* off = reinterpret_cast<unsigned*>(win_w + ((MASK & 2) ? wcells : 0));
    float* sv = reinterpret_cast<float*>(off + ((cells + 1 + 3) & ~3));
    float* ssx = sv + P.cap;
    float* ssy = ssx + P.cap;
    __shared__ unsigned wave_tot[kTileThreads / 64];
    __shared__ int next_row;

    // ---- the item's records: loads first, then the LDS set-up they hide behind
    const uint4* rec = records + it.first;
    uint4 rc[kRecPerThread];
#pragma unroll
    for (int k = 0; k < kRecPerThread; ++k) {
        const unsigned j = threadIdx.x + k * kTileThreads;
        rc[k] = j < it.count ? stream_load(rec + j) : make_uint4(kNullCell, 0u, 0u, 0u);
        // a record is only ever trusted as far as the tile goes: a local cell outside it (a record written for another tile
        // geometry, a slot the scatter pass never filled) is dropped here instead of indexing the LDS arrays below
        if (rc[k].x >= (unsigned)cells) rc[k].x = kNullCell;
    }
    for (int i = threadIdx.x; i <= cells; i += kTileThreads) off[i] = 0;
    {
        const int nwin = wcells * (((MASK & 1) ? 1 : 0) + ((MASK & 2) ? 1 : 0));
        for (int i = threadIdx.x; i < nwin; i += kTileThreads) win_s[i] = 0.0;
    }
    if (threadIdx.x == 0) next_row = 0;
    __syncthreads();

    // ---- counting sort by cell: rank from the counting atomic, in-place exclusive scan, staged SoA
    unsigned rk[kRecPerThread];
#pragma unroll
    for (int k = 0; k < kRecPerThread; ++k) {
        rk[k] = 0;
        if (rc[k].x != kNullCell) rk[k] = atomicAdd(&off[rc[k].x], 1u);
    }
    __syncthreads();
    {
        constexpr int kMaxPer = 8;                                             // cells <= 8 * THREADS (checked on the host)
        const int per = (cells + kTileThreads - 1) / kTileThreads;
        const int lo = min((int)threadIdx.x * per, cells), hi = min(lo + per, cells);
        unsigned c[kMaxPer], s = 0;
#pragma unroll
        for (int q = 0; q < kMaxPer; ++q) {
            c[q] = 0u;
            if (q < per && lo + q < hi) { c[q] = off[lo + q]; s += c[q]; }
        }
        unsigned incl = s;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        unsigned run = incl - s;
        for (int wv = 0; wv < wave; ++wv) run += wave_tot[wv];
#pragma unroll
        for (int q = 0; q < kMaxPer; ++q)
            if (q < per && lo + q < hi) { off[lo + q] = run; run += c[q]; }
        if (threadIdx.x == kTileThreads - 1) off[cells] = run;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRecPerThread; ++k) {
        if (rc[k].x == kNullCell) continue;
        const unsigned pos = off[rc[k].x] + rk[k];
        sv[pos] = __uint_as_float(rc[k].y);
        ssx[pos] = __uint_as_float(rc[k].z);
        ssy[pos] = __uint_as_float(rc[k].w);
    }
    __syncthreads();

    // ---- rows of cells: a wave takes a row at a time, lane = cell
    const int lane = threadIdx.x & 63;
    const int bx = it.bin % b.bins_x, by = it.bin / b.bins_x;
    const int col0 = bx * b.tile_w;                               // global column of the tile's first cell
    const int row0 = g.st_r0 + b.row0 + by * b.tile_h;            // global row of the tile's first row
    for (;;) {
        int row = 0;
        if (lane == 0) row = atomicAdd(&next_row, 1);
        row = __builtin_amdgcn_readfirstlane(row);
        if (row >= b.tile_h) break;
        const bool active = lane < b.tile_w;
        const int c = row * b.tile_w + lane;
        const unsigned e0 = active ? off[c] : 0u;
        const unsigned cnt = active ? min(off[c + 1] - e0, (unsigned)P.cap) : 0u;      // (the bound only guards the loop below)
        if (!__any(cnt > 0)) continue;

        // clip rectangle of the centre cell (its reference tile and the state window, Q4): every point of the cell shares it
        const int gc = col0 + lane, gr = row0 + row;
        const int tcx = fast_div(min(gc, g.W - 1), g.tw), tcy = fast_div(min(gr, g.H - 1), g.th);
        const int cx0 = tcx * g.tw, cx1 = min(cx0 + g.tw, g.W);
        const int cy0 = max(tcy * g.th, g.st_r0), cy1 = min(min(tcy * g.th + g.th, g.H), g.st_r0 + g.st_rows);
        const int jlo = cx0 - gc + R, jhi = cx1 - 1 - gc + R;             // valid j in [jlo, jhi]
        const int ilo = cy0 - gr + R, ihi = cy1 - 1 - gr + R;
        const bool clipped = cnt > 0 && (jlo > 0 || jhi < D - 1 || ilo > 0 || ihi < D - 1);
        const bool any_clipped = __any(clipped);

        // acc[i][j] = {sum of v w, sum of w} of footprint cell (i, j), both planes in one register pair: one packed fused
        // multiply-add per cell and point.  The kernel is bound by vector-instruction ISSUE -- a wave64 instruction holds
        // its SIMD for four cycles, packed or not (rocprofv3: 6.9e8 instructions, SIMDs 95 % busy at 1.38 ms,
        // profiles/r03_gauss1_sq.md) -- so the two planes ride on v_pk_fma_f32 (250 -> ~125 instructions per point).
        pcr_f2 acc[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) acc[i][j] = pcr_f2{0.f, 0.f};
        // Every lane runs every round: a lane whose cell has fewer points folds a NULL point (value 0, x weights 0), so the
        // body is straight-line code (predicated, every accumulator went through a select per round).
        for (unsigned k = 0; __any(k < cnt); ++k) {
            const bool live = k < cnt;
            const unsigned idx = live ? e0 + k : 0u;
            const float val = live ? sv[idx] : 0.f;
            float ex[D], ey[D];
            axis_weights<R>(ssx[idx], P.inv_s2x, P.gx, ex, live ? 1.f : 0.f);
            axis_weights<R>(ssy[idx], P.inv_s2y, P.gy, ey, 1.f);
            pcr_f2 vx[D];                                        // {v wx[j], wx[j]}
#pragma unroll
            for (int j = 0; j < D; ++j) vx[j] = pcr_f2{val, 1.0f} * pcr_f2{ex[j], ex[j]};
#pragma unroll
            for (int i = 0; i < D; ++i) {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const bool corner = (i == 0 || i == D - 1) && (j == 0 || j == D - 1);
                    float wy = ey[i];
                    if (CUT == 2 || (CUT == 1 && corner)) wy = (ex[j] * ey[i] < 1e-6f) ? 0.f : wy;       // glyph_kernels.cu:166
                    acc[i][j] = __builtin_elementwise_fma(vx[j], pcr_f2{wy, wy}, acc[i][j]);
                }
            }
        }
        if (any_clipped) {
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j)
                    if (j < jlo || j > jhi || i < ilo || i > ihi) acc[i][j] = pcr_f2{0.f, 0.f};
        }
        // columns of neighbouring lanes' footprints meet: after 2R shifts lane l holds window column l of footprint row i
        float ts[D], tw[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            ts[i] = acc[i][D - 1].x;
            tw[i] = acc[i][D - 1].y;
#pragma unroll
            for (int j = D - 2; j >= 0; --j) {
                if (MASK & 1) ts[i] = wave_shr1(ts[i]) + acc[i][j].x;
                if (MASK & 2) tw[i] = wave_shr1(tw[i]) + acc[i][j].y;
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const int wi = (row + i) * 64 + lane;                // window row 0 = tile row -R
            if ((MASK & 1) && ts[i] != 0.f) unsafeAtomicAdd(&win_s[wi], (double)ts[i]);
            if ((MASK & 2) && tw[i] != 0.f) unsafeAtomicAdd(&win_w[wi], (double)tw[i]);
        }
    }
    __syncthreads();

    // ---- merge: window -> planes, float atomics on contiguous row segments (windows of neighbouring tiles overlap)
    for (int i = threadIdx.x; i < wcells; i += kTileThreads) {
        const double s = (MASK & 1) ? win_s[i] : 0.0;
        const double w = (MASK & 2) ? win_w[i] : 0.0;
        if (s == 0.0 && w == 0.0) continue;
        const int wr = i >> 6, wc = i & 63;
        const int64_t cell = (int64_t)(row0 - R + wr - g.st_r0) * g.W + (col0 - R + wc);     // non-zero cells were clipped already
        if ((MASK & 1) && s != 0.0) atomic_add_f32(pl.sum + cell, (float)s);
        if ((MASK & 2) && w != 0.0) atomic_add_f32(pl.wgt + cell, (float)w);
    }
}

// ---- host ----------------------------------------------------------------------------------------------------
struct CellPlan {
    int R, cut, threads;
    CellGauss P;
    int tile_w, tile_h;
    size_t lds;
};

// tile height and item size from the LDS: window (64 x (h + 2R) x 8 B per plane) + offsets + 12 B per staged record
bool plan_cells(const pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask, CellPlan* out) {
    const GridDev& g = e->gd;
    if (gl.type != PCR_HIP_GLYPH_GAUSSIAN || gl.sigma_x || gl.sigma_y || gl.rotation || gl.def_rotation != 0.0f) return false;
    const float sx = gl.def_sigma_x * (float)g.inv_csx, sy = gl.def_sigma_y * (float)g.inv_csy;
    const float Rf = std::fmin(3.0f * std::fmax(sx, sy), gl.max_radius);       // gauss_params (glyph_device.hpp)
    if (!(Rf == Rf)) return false;
    const int r = std::min((int)std::ceil(Rf), 1 << 20);
    if (r < 1 || r > 3) return false;
    if (!(sx * sx > 0.f) || !(sy * sy > 0.f) || !std::isfinite(1.0f / (sx * sx)) || !std::isfinite(1.0f / (sy * sy))) return false;
    // exp(u / s^2) up to the R-th power must stay far from the f32 range: s^2 >= 1/16 (the radius rule gives s <= R / 3)
    if (sx * sx < 0.0625f || sy * sy < 0.0625f) return false;
    CellPlan p;
    p.R = r;
    p.P.inv_s2x = 1.0f / (sx * sx);
    p.P.inv_s2y = 1.0f / (sy * sy);
    for (int d = 0; d < 4; ++d) {
        p.P.gx[d] = (float)std::exp(-(double)d * d / (2.0 * (double)sx * sx));
        p.P.gy[d] = (float)std::exp(-(double)d * d / (2.0 * (double)sy * sy));
    }
    // which footprint cells can fall under 1e-6: the largest exponent of cell (i, j) over the sub-cell offsets u in [0, 1)
    auto worst = [&](int d, double s) { const double a = std::max(std::fabs((double)d), std::fabs((double)d - 1.0)); return a * a / (2.0 * s * s); };
    const double limit = 13.8155 * 0.999;                                         // -ln(1e-6), with a margin for the f32 arithmetic
    bool any_inner = false, any_corner = false;
    for (int i = -r; i <= r; ++i)
        for (int j = -r; j <= r; ++j) {
            if (worst(j, sx) + worst(i, sy) < limit) continue;
            if (std::abs(i) == r && std::abs(j) == r) any_corner = true;
            else any_inner = true;
        }
    p.cut = any_inner ? 2 : (any_corner ? 1 : 0);
    p.tile_w = 64 - 2 * r;
    const int planes = ((mask & 1) ? 1 : 0) + ((mask & 2) ? 1 : 0);
    // Tile height: 20 rows -> ~3.5 K records at three per cell, < 80 KB of LDS: TWO 512-thread workgroups per CU, one's
    // record loads, sort and merge run behind the other's arithmetic.  (One 1024-thread workgroup on 40-row tiles: the
    // phases of a tile run back to back, k_cell_gauss 1.46 ms instead of ...; PCR_HIP_CELL_TILE_H = 40 selects it.)
    // A tile with more records than the LDS holds is split into several work items.
    int best_h = 20;
    if (const char* t = std::getenv("PCR_HIP_CELL_TILE_H")) {                      // experiments: force the tile height
        const int h = std::atoi(t);
        if (h >= 8 && h <= 48 && h % 4 == 0) best_h = h;
    } else if ((int64_t)((g.W + p.tile_w - 1) / p.tile_w) * ((g.st_rows + best_h - 1) / best_h) > b16::max_bins(e) &&
               e->max_bins == kMaxBins) {
        best_h = 40;            // a window with more 20-row tiles than one binning pass takes (a C5 shard): one pass of 40-row tiles
    }
    const size_t budget = best_h <= 24 ? (size_t)80 * 1024 - 1024 : (size_t)160 * 1024 - 2048;
    const size_t fixed = (size_t)64 * (best_h + 2 * r) * 8 * planes + (((size_t)p.tile_w * best_h + 1 + 3) & ~size_t(3)) * 4;
    if (fixed + 12 * 1024 > budget) return false;
    // Workgroup shape.  R = 3 keeps 98 accumulator registers per lane (~215 VGPRs: two waves per SIMD), i.e. eight
    // waves per CU: two 256-thread workgroups on 20-row tiles, or one 512-thread workgroup on 40-row tiles.  (Two
    // column passes over the cell's points -- 56 + 42 accumulators, <= 128 VGPRs, sixteen waves per CU -- cost 1.9x the
    // arithmetic of a kernel that is issue-bound: 1.89 vs 1.46 ms.  Removed.)
    const bool heavy = r == 3;
    if (best_h <= 24) p.threads = heavy ? 256 : 512;
    else p.threads = heavy ? 512 : 1024;
    const int best_cap = std::min((int)((budget - fixed) / 12) & ~255, (p.threads == 1024 ? 8 : 16) * p.threads);
    if (p.tile_w * best_h > 8 * p.threads) return false;           // the tile kernel's in-LDS scan: <= 8 cells per thread
    p.tile_h = best_h;
    p.P.cap = best_cap;
    p.lds = (size_t)64 * (best_h + 2 * r) * 8 * planes + (((size_t)p.tile_w * best_h + 1 + 3) & ~size_t(3)) * 4 + (size_t)best_cap * 12;
    // The bounds every LDS index of k_cell_gauss rests on (the r03c aperture violation, DESIGN section 9, was a launch of
    // this kernel's 512-thread, 40-row shape from an uncommitted experiment; the record names no address, so every bound
    // is pinned here instead of being implied):
    //   an item holds <= cap records and the kernel keeps <= kRecPerThread x THREADS of them in registers;
    //   the in-LDS scan walks <= 8 cells per thread; the carve-up ends inside the CU's 160 KB with the static words.
    const int rec_per_thread = p.threads == 1024 ? 8 : 16;
    if (best_cap <= 0 || best_cap > rec_per_thread * p.threads) return false;
    if (p.lds + 64 > (size_t)160 * 1024 || fixed + (size_t)best_cap * 12 != p.lds) return false;
    *out = p;
    return true;
}

BinGeom cell_bins(const GridDev& g, const CellPlan& p, int row0, int rows) {
    BinGeom b;
    b.tile_w = p.tile_w;
    b.tile_h = p.tile_h;
    b.bins_x = (g.W + b.tile_w - 1) / b.tile_w;
    b.bins_y = (rows + b.tile_h - 1) / b.tile_h;
    b.nbins = b.bins_x * b.bins_y;
    b.chunk = 0;                                         // set by the caller (the engine picks the scatter shape)
    b.row0 = row0;
    b.rows = rows;
    b.sup_shift = 0;
    return b;
}

template <int R, unsigned MASK>
void launch_cells(pcr_hip_engine* e, const GridDev& gd, const BinGeom& b, const CellPlan& p, const PlanesDev& pl,
                  const uint4* rec, const BinItem* items, const unsigned* n_items, int max_items) {
    auto go = [&](auto kernel, int threads) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
        hipLaunchKernelGGL(kernel, dim3(max_items), dim3(threads), p.lds, e->stream, gd, b, p.P, pl, rec, items, n_items);
    };
    auto by_cut = [&](auto threads_c) {
        constexpr int T = decltype(threads_c)::value;
        if (p.cut == 0) go(&k_cell_gauss<R, MASK, 0, T>, T);
        else if (p.cut == 1) go(&k_cell_gauss<R, MASK, 1, T>, T);
        else go(&k_cell_gauss<R, MASK, 2, T>, T);
    };
    using std::integral_constant;
    if constexpr (R == 3) {
        if (p.threads == 256) by_cut(integral_constant<int, 256>{});
        else by_cut(integral_constant<int, 512>{});
    } else {
        if (p.threads == 512) by_cut(integral_constant<int, 512>{});
        else by_cut(integral_constant<int, 1024>{});
    }
}

template <unsigned MASK>
void launch_cells_r(pcr_hip_engine* e, const GridDev& gd, const BinGeom& b, const CellPlan& p, const PlanesDev& pl,
                    const uint4* rec, const BinItem* items, const unsigned* n_items, int max_items) {
    if (p.R == 1) launch_cells<1, MASK>(e, gd, b, p, pl, rec, items, n_items, max_items);
    else if (p.R == 2) launch_cells<2, MASK>(e, gd, b, p, pl, rec, items, n_items, max_items);
    else launch_cells<3, MASK>(e, gd, b, p, pl, rec, items, n_items, max_items);
}

}  // namespace

namespace pcrhip {

bool cells_gauss_supported(const pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask) {
    if (mask == 0 || (mask & ~3u)) return false;
    CellPlan p;
    if (!plan_cells(e, gl, mask, &p)) return false;
    const GridDev& g = e->gd;
    const int bins_x = (g.W + p.tile_w - 1) / p.tile_w;
    const int band_rows = band_rows_for(g, p.tile_w, p.tile_h, b16::max_bins(e));
    if (band_rows <= 0) return false;
    // Every band is a full pass over the points (count + scatter: ~12 us per million points and band), the cell tiles
    // themselves cost ~28 us per million points, the index-record tiles behind the two-level sort ~83 (the full-size two-rank
    // rehearsals, profiles/r04_c5_rehearsal_2ranks_one_gpu.json: a 16384 x 8192 shard, four bands, 75.7 against 82.9 ms per
    // 500 M points with both ranks on one GPU): the sweep wins up to four bands.  Round 3 stopped at two.
    (void)bins_x;
    const int nbands = (g.st_rows + band_rows - 1) / band_rows;
    if (nbands > (e->max_bins == kMaxBins ? 4 : kMaxBands)) return false;
    // every tile's window is swept once per scatter: not worth it for a handful of points
    const uint64_t cells = (uint64_t)g.W * g.st_rows;
    if (e->forced_path != 2 && e->stats.points_in * 64 < cells) return false;
    return e->stats.points_in < (1ull << 32) - (1ull << 20);
}

int cells_gauss(pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask, const PlanesDev& pl,
                const double* x, const double* y, const float* v, uint64_t n) {
    CellPlan p;
    if (!plan_cells(e, gl, mask, &p)) return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_glyph: cell tiles not applicable");
    const GridDev& ge = e->gd;
    const int band_rows = band_rows_for(ge, p.tile_w, p.tile_h, b16::max_bins(e));
    if (band_rows <= 0) return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_glyph: grid cannot be binned");
    const int max_bins = ((ge.W + p.tile_w - 1) / p.tile_w) * ((std::min(band_rows, ge.st_rows) + p.tile_h - 1) / p.tile_h);
    const unsigned item_records = (unsigned)p.P.cap;
    const Layout L = layout(0, max_bins, n, item_records);
    int rc = ensure_scratch(e, L.end);
    if (rc) return rc;
    Buffers bb{};
    PCR_HIP_TRY(hipMemsetAsync(e->d_scratch + L.o_fbc, 0, 4, e->stream));
    int total_bins = 0;
    for (int row0 = 0; row0 < ge.st_rows; row0 += band_rows) {
        const int rows = std::min(band_rows, ge.st_rows - row0);
        GridDev gd = ge;                                          // this band's points only
        gd.own_r0 = std::max(ge.own_r0, ge.st_r0 + row0);
        gd.own_r1 = std::min(ge.own_r1, ge.st_r0 + row0 + rows);
        if (gd.own_r0 >= gd.own_r1) continue;
        BinGeom b = cell_bins(ge, p, row0, rows);
        b.chunk = chunk_of<GaussCellMaker>();
        total_bins += b.nbins;
        rc = bin(e, gd, b, GaussCellMaker{}, x, y, v, n, item_records, L, &bb);
        if (rc) return rc;
        {
            // the tile kernel re-derives clip rectangles from the engine's grid: the band only selected the points
            ScopedKernelTimer t(e, "k_cell_gauss");
            if (mask == 1) launch_cells_r<1>(e, ge, b, p, pl, bb.records, bb.items, bb.n_items, bb.max_items);
            else if (mask == 2) launch_cells_r<2>(e, ge, b, p, pl, bb.records, bb.items, bb.n_items, bb.max_items);
            else launch_cells_r<3>(e, ge, b, p, pl, bb.records, bb.items, bb.n_items, bb.max_items);
        }
    }
    unsigned* d_fbl = reinterpret_cast<unsigned*>(e->d_scratch + L.o_fbl);
    unsigned* d_fbc = reinterpret_cast<unsigned*>(e->d_scratch + L.o_fbc);
    {
        // points the cell form cannot represent: painted directly, on the engine's own grid and planes
        ScopedKernelTimer t(e, "k_gauss_list");
        if (mask == 1) hipLaunchKernelGGL(k_cell_gauss_list<1>, dim3(64), dim3(256), 0, e->stream, ge, gl, pl, d_fbl, d_fbc, x, y, v);
        else if (mask == 2) hipLaunchKernelGGL(k_cell_gauss_list<2>, dim3(64), dim3(256), 0, e->stream, ge, gl, pl, d_fbl, d_fbc, x, y, v);
        else hipLaunchKernelGGL(k_cell_gauss_list<3>, dim3(64), dim3(256), 0, e->stream, ge, gl, pl, d_fbl, d_fbc, x, y, v);
    }
    PCR_HIP_TRY(hipGetLastError());
    e->stats.path = 1;
    e->stats.lds_tile_w = p.tile_w;
    e->stats.lds_tile_h = p.tile_h;
    e->stats.lds_apron = p.R;
    e->stats.num_bins = total_bins;
    return PCR_HIP_OK;
}

}  // namespace pcrhip
