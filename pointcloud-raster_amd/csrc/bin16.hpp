// bin16.hpp -- the binning front-end on 16-byte VALUE records shared by the glyph tile kernels
// (scatter_cells.hip: Gaussian cell tiles; scatter_binned_glyph.hip: Line tiles; scatter_moments.hip: moment tiles).
//
//   k_b16_count    x, y            -> counts per (virtual XCD, tile); touched tiles; the list of points the record cannot hold
//   k_b16_scan     counts          -> sub-range starts, tile starts, work items (a crowded tile is split)
//   k_b16_scatter  x, y, v, chans  -> finished 16-byte records, stored straight from registers
//
// No routing keys travel between the passes: the scatter pass has to read x, y anyway (a record carries what the
// footprint needs from them) and routes again, bit for bit as the counting pass did.  52 bytes per point (16 + 20 + 16)
// where round 2's forms moved 64 (moments) to 76 (Line: keys + 32-byte records).  Small tiles make a (workgroup, tile)
// run one or two records long, so nothing is staged in LDS; instead every tile's record range is split into EIGHT
// sub-ranges, one per VIRTUAL XCD (blockIdx % 8 in both passes; workgroups are dealt to the XCDs round-robin), so that
// a 128-byte line of records is only ever written through one XCD's L2 and leaves it whole.  The mapping is a speed
// matter only: whatever the placement, the same records land in the same tile.
//
// A Maker turns a routed point into its record:
//     static constexpr int kPer, kBatch;                                points per thread of the scatter pass (x threads = chunk);
//                                                                       how many of them are loaded before the first is routed
//     struct Chan;                                                      per-point channel values it needs
//     Chan load(uint64_t i) const;                                      issued with the x, y, v loads of a batch
//     bool make(g, b, routed, pg, value, chan, uint4& rec) const;       fills rec.y/.z/.w; false: the record cannot hold
//                                                                       this point -> null record + the list
//     static constexpr bool kOwnsX;                                     false: .x = the local centre cell (set by the pass);
//                                                                       true: make() fills .x as well (kNullCell = nothing to do)
//     static constexpr bool kFixup;                                     true: make() may return kLater (2) -- "this point needs
//                                                                       the long way" -- and the Maker has
//     bool fixup(g, b, routed, pg, i, value, uint4& rec) const;         which the pass calls for those points AFTER its main
//                                                                       loop, compacted over the workgroup (below)
// Replaces tile_router_assign_gpu + tile_router_sort_gpu (src/engine/tile_router_kernels.cu:34-293) for glyph clouds.
#pragma once

#include "engine.hpp"
#include "glyph_device.hpp"

namespace pcrhip {
namespace b16 {

constexpr int kCountThreads = 512;
constexpr int kScatThreads = 512;                         // scatter pass: two workgroups per CU (see chunk_of below)
constexpr int kVx = 8;                                     // virtual XCDs
constexpr unsigned kNullCell = 0xFFFFFFFFu;
constexpr int kLater = 2;                                  // Maker::make of a kFixup Maker: handle this point after the main loop
constexpr int kItemMax = 8192;                             // tile kernel: records per work item (held in registers while they are ranked)

__device__ __forceinline__ bool finite_f(float v) { return (__float_as_uint(v) & 0x7F800000u) != 0x7F800000u; }

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// The bin geometry in vector registers too (see vector_resident in common.hpp): the kernels below route with it per point.
__device__ __forceinline__ BinGeom vector_resident(const BinGeom& u) {
    BinGeom b = u;
    b.tile_w = pcrhip::vector_resident(u.tile_w); b.tile_h = pcrhip::vector_resident(u.tile_h);
    b.bins_x = pcrhip::vector_resident(u.bins_x); b.row0 = pcrhip::vector_resident(u.row0);
    return b;
}

// ---- classification shared by both passes ------------------------------------------------------------------
// 0: not this band's point; 1: binned (bin, lcell valid); 2: valid but not representable by geometry -> list
struct Routed16 {
    int kind;
    int bin;
    unsigned lcell;
    int bx, by;                 // the tile's column / row (bin = by * bins_x + bx): a Maker that needs the tile's origin
};

// CENTRE: the record's footprint is positioned by floor(fc), which must then be the routed cell (Gaussians; a Line
// record carries its own end points).
template <bool CENTRE>
__device__ __forceinline__ Routed16 classify(const GridDev& g, const BinGeom& b, uint64_t i, double wx, double wy, PointGeom& pg) {
    Routed16 r{0, 0, 0u, 0, 0};
    pg = point_geom(g, wx, wy);
    if (!(pg.valid && point_kept(g, i))) return r;
    if (CENTRE) {
        const int icx = (int)floor(pg.fcx), icy = (int)floor(pg.fcy);
        if (icx != pg.col || icy != pg.row) { r.kind = 2; return r; }   // centre of the footprint != routed cell (grid edge)
    }
    const int sr = pg.row - g.st_r0 - b.row0;
    const int bx = fast_div(pg.col, b.tile_w), by = fast_div(sr, b.tile_h);
    r.kind = 1;
    r.bx = bx;
    r.by = by;
    r.bin = by * b.bins_x + bx;
    r.lcell = (unsigned)((sr - by * b.tile_h) * b.tile_w + (pg.col - bx * b.tile_w));
    return r;
}

// ---- pass A: counts per (virtual XCD, tile); x, y only ---------------------------------------------------------
template <bool CENTRE>
__global__ void __launch_bounds__(kCountThreads)
k_b16_count(GridDev g_uniform, BinGeom b_uniform, int cb, const double* __restrict__ x, const double* __restrict__ y, uint64_t n,
            unsigned* __restrict__ cnt, unsigned* __restrict__ fb_list, unsigned* __restrict__ fb_count,
            uint32_t* __restrict__ touched, unsigned long long* __restrict__ counters) {
    const GridDev g = pcrhip::vector_resident<PCR_VRES_B16_COUNT>(g_uniform);      // (common.hpp: the scalar registers do not hold all of it)
    const BinGeom b = PCR_VRES_B16_COUNT >= 2 ? vector_resident(b_uniform) : b_uniform;
    extern __shared__ unsigned lds_hist[];
    for (int i = threadIdx.x; i < b.nbins; i += kCountThreads) lds_hist[i] = 0;
    __shared__ unsigned any_valid;
    __shared__ unsigned lds_touch[kTouchLdsTiles];
    TouchLds tl;
    tl.begin(g, lds_touch, kCountThreads);
    if (threadIdx.x == 0) any_valid = 0;
    __syncthreads();
    const bool one_tile = g.tiles_x * g.tiles_y == 1;
    // A workgroup counts `cb` chunks of ONE virtual XCD (chunks vx, vx + 8, vx + 16, ...): its histogram is flushed with up
    // to nbins global atomics, once per cb * chunk points instead of once per chunk.
    const int vx = blockIdx.x & (kVx - 1);
    const uint64_t group = blockIdx.x / kVx;
    unsigned my_valid = 0;
    auto handle = [&](uint64_t i, double wx, double wy) {
        PointGeom pg;
        const Routed16 r = classify<CENTRE>(g, b, i, wx, wy, pg);
        if (r.kind == 0) return;
        ++my_valid;
        if (!one_tile) tl.touch(g, touched, pg.row, pg.col);
        if (r.kind == 1) atomicAdd(&lds_hist[r.bin], 1u);
        else fb_list[atomicAdd(fb_count, 1u)] = (unsigned)i;
    };
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    for (int t = 0; t < cb; ++t) {
        const uint64_t base = ((group * cb + t) * kVx + vx) * (uint64_t)b.chunk;   // b.chunk: the scatter pass's chunk, a multiple of 1024
        if (base >= n) break;
        if (aligned && base + (uint64_t)b.chunk <= n) {
            const double2* x2 = reinterpret_cast<const double2*>(x + base);
            const double2* y2 = reinterpret_cast<const double2*>(y + base);
            const int pairs = b.chunk >> 1;
            for (int p0 = threadIdx.x; p0 < pairs; p0 += 4 * kCountThreads) {
                double2 xs[4], ys[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (p0 + u * kCountThreads >= pairs) break;              // uniform
                    xs[u] = stream_load(x2 + p0 + u * kCountThreads);
                    ys[u] = stream_load(y2 + p0 + u * kCountThreads);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (p0 + u * kCountThreads >= pairs) break;
                    const uint64_t i = base + 2ull * (p0 + u * kCountThreads);
                    handle(i, xs[u].x, ys[u].x);
                    handle(i + 1, xs[u].y, ys[u].y);
                }
            }
        } else {
            for (int k = threadIdx.x; k < b.chunk; k += kCountThreads) {
                const uint64_t i = base + k;
                if (i >= n) break;
                handle(i, x[i], y[i]);
            }
        }
    }
    if (my_valid) atomicAdd(&any_valid, my_valid);
    __syncthreads();
    unsigned* mine = cnt + (size_t)vx * b.nbins;
    for (int i = threadIdx.x; i < b.nbins; i += kCountThreads) {
        const unsigned c = lds_hist[i];
        if (c) atomicAdd(&mine[i], c);
    }
    tl.flush(g, touched, kCountThreads);
    if (threadIdx.x == 0 && any_valid) {
        atomicAdd(counters, (unsigned long long)any_valid);
        if (one_tile) touched[0] = 1u;
    }
}

// ---- scan: sub-range starts per (virtual XCD, tile), tile starts, work items ------------------------------------
// A tile's records are laid out [vx 0 | vx 1 | ... | vx 7]; an item is at most item_records of one tile.
// One tile per thread, 1024 tiles per workgroup (coalesced: counts are laid out [vx][tile]); a workgroup sums the tiles of
// the workgroups before it by itself -- at most 15 x 8 coalesced loads per thread at 16384 tiles, cheaper than a second
// launch.  (The first version walked `per` consecutive tiles per thread in ONE workgroup: 0.2 ms at 16384 tiles.)
static __global__ void __launch_bounds__(1024)      // (static: the header is compiled into three translation units)
k_b16_scan(int nbins, unsigned item_records, const unsigned* __restrict__ cnt, unsigned* __restrict__ cursor,
           unsigned* __restrict__ bin_start, BinItem* __restrict__ items, unsigned* __restrict__ n_items) {
    __shared__ unsigned part[1024];
    __shared__ unsigned ipart[1024];
    const int tid = threadIdx.x;
    auto tile_total = [&](int i, unsigned (&cv)[kVx]) {
        unsigned c = 0;
#pragma unroll
        for (int v = 0; v < kVx; ++v) { cv[v] = cnt[(size_t)v * nbins + i]; c += cv[v]; }
        return c;
    };
    // records and items of every tile before this workgroup's first
    unsigned s_prev = 0, it_prev = 0;
    for (int pb = 0; pb < (int)blockIdx.x; ++pb) {
        unsigned cv[kVx];
        const unsigned c = tile_total(pb * 1024 + tid, cv);
        s_prev += c;
        it_prev += (c + item_records - 1) / item_records;
    }
    part[tid] = s_prev;
    ipart[tid] = it_prev;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if (tid < off) { part[tid] += part[tid + off]; ipart[tid] += ipart[tid + off]; }
        __syncthreads();
    }
    const unsigned base = part[0], ibase = ipart[0];
    __syncthreads();
    // this workgroup's tiles
    const int i = (int)blockIdx.x * 1024 + tid;
    unsigned cv[kVx] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, c = 0, it = 0;
    if (i < nbins) {
        c = tile_total(i, cv);
        it = (c + item_records - 1) / item_records;
    }
    part[tid] = c;
    ipart[tid] = it;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned a = 0, c2 = 0;
        if (tid >= off) { a = part[tid - off]; c2 = ipart[tid - off]; }
        __syncthreads();
        part[tid] += a;
        ipart[tid] += c2;
        __syncthreads();
    }
    if (i < nbins) {
        const unsigned run = base + part[tid] - c, irun = ibase + ipart[tid] - it;
        unsigned acc = 0;
#pragma unroll
        for (int v = 0; v < kVx; ++v) {
            cursor[(size_t)v * nbins + i] = run + acc;
            acc += cv[v];
        }
        bin_start[i] = run;                                   // [nbins + 1]: for tile kernels that walk a whole tile
        for (unsigned p = 0; p < it; ++p)
            items[irun + p] = BinItem{(unsigned)i, run + p * item_records, min(item_records, c - p * item_records), it > 1 ? 1u : 0u};
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 1023) {
        *n_items = ibase + ipart[1023];
        bin_start[nbins] = base + part[1023];
    }
}

// ---- pass B: records straight from registers -------------------------------------------------------------------
template <class Maker>
__global__ void __launch_bounds__(kScatThreads, 4)     // <= 128 VGPRs: two 512-thread workgroups per CU
k_b16_scatter(GridDev g_uniform, BinGeom b_uniform, Maker mk, const double* __restrict__ x, const double* __restrict__ y,
              const float* __restrict__ v, uint64_t n, unsigned* __restrict__ cursor, uint4* __restrict__ records,
              unsigned* __restrict__ fb_list, unsigned* __restrict__ fb_count) {
    // (common.hpp: the scalar registers do not hold all of it -- for the Makers that have the vector registers to spare)
    const GridDev g = Maker::kVectorGeometry ? pcrhip::vector_resident<PCR_VRES_B16_SCATTER>(g_uniform) : g_uniform;
    const BinGeom b = Maker::kVectorGeometry && PCR_VRES_B16_SCATTER >= 2 ? vector_resident(b_uniform) : b_uniform;
    extern __shared__ unsigned lds_hist[];                  // [nbins]: rank counters, then the run's global start
    // Makers with a rare long way (kFixup): the points that need it are LISTED in LDS by the unrolled main loop -- which then
    // holds the short way only -- and handled afterwards by as many lanes as there are such points.  (A rare per-lane branch
    // inside the unrolled loop is taken by two wave iterations in three when 2 % of the points need it, each time for one or
    // two lanes, and its code is there once per unrolled point: the Line pass had grown to 14 000 instructions and 361
    // executed vector instructions per point; with the list 8 200 and 308, A/B in one call 0.67 -> 0.63 ms.)
    uint2* fix_list = reinterpret_cast<uint2*>(lds_hist + ((b.nbins + 3) & ~3));      // {point offset in the chunk, rank | bin << 16}
    __shared__ unsigned fix_count;
    if (threadIdx.x == 0) fix_count = 0;
    for (int i = threadIdx.x; i < b.nbins; i += kScatThreads) lds_hist[i] = 0;
    __syncthreads();
    constexpr int kScatPer = Maker::kPer;                    // points per thread: what the Maker's arithmetic leaves registers for
    const uint64_t base = (uint64_t)blockIdx.x * (kScatThreads * kScatPer);
    uint4 rec[kScatPer];                                     // .x = bin << 16 | local cell until the store
    unsigned rank[kScatPer];
    // every load of the chunk is issued before the first point is routed: one memory latency per workgroup instead of
    // one per batch (72 % of this kernel's wave-cycles were waits, profiles/r03_gauss1_sq.md); a routed point's record
    // takes the registers its x, y, value came in
    constexpr int kBatch = Maker::kBatch;
#pragma unroll
    for (int k0 = 0; k0 < kScatPer; k0 += kBatch) {
        double wx[kBatch], wy[kBatch];
        float val[kBatch];
        typename Maker::Chan ch[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint64_t i = base + (uint64_t)(k0 + u) * kScatThreads + threadIdx.x;
            const uint64_t ic = i < n ? i : n - 1;
            // read once here: streamed past the caches, which hold the record lines being filled
            wx[u] = __builtin_nontemporal_load(x + ic);
            wy[u] = __builtin_nontemporal_load(y + ic);
            val[u] = __builtin_nontemporal_load(v + ic);
            ch[u] = mk.load(ic);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int k = k0 + u;
            const uint64_t i = base + (uint64_t)k * kScatThreads + threadIdx.x;
            rec[k] = make_uint4(kNullCell, 0u, 0u, 0u);
            rank[k] = 0xFFFFFFFFu;                           // no slot
            if (i >= n) continue;
            PointGeom pg;
            const Routed16 r = classify<Maker::kCentre>(g, b, i, wx[u], wy[u], pg);
            if (r.kind != 1) continue;                       // kind 2 was listed by the counting pass
            rank[k] = atomicAdd(&lds_hist[r.bin], 1u) | ((unsigned)r.bin << 16);          // rank < 2^14 (chunk), bin < 2^16
            if constexpr (Maker::kFixup) {
                static_assert(Maker::kOwnsX, "a kFixup Maker fills the whole record, .x included, on both ways");
                const int code = mk.make(g, b, r, pg, val[u], ch[u], rec[k]);             // 1 done, 0 list, kLater: the long way
                if (code == 0) fb_list[atomicAdd(fb_count, 1u)] = (unsigned)i;
                else if (code == kLater) {
                    fix_list[atomicAdd(&fix_count, 1u)] = make_uint2((unsigned)(i - base), rank[k]);
                    rec[k].x = kNullCell;                                                   // the slot holds a null record until then
                }
            } else {
                if (mk.make(g, b, r, pg, val[u], ch[u], rec[k])) { if (!Maker::kOwnsX) rec[k].x = r.lcell; }   // (kOwnsX: the Maker filled .x itself)
                else fb_list[atomicAdd(fb_count, 1u)] = (unsigned)i;                        // the slot keeps a null record
            }
        }
    }
    __syncthreads();
    {
        unsigned* mine = cursor + (size_t)(blockIdx.x & (kVx - 1)) * b.nbins;
        // a lane's reservations are issued back to back, eight at a time: a returning global atomic takes microseconds
        // under load, and at 16 384 tiles a lane owns 32 of them -- eight rounds of four were ~24 us of pure latency per
        // workgroup, more than its loads and stores together
        constexpr int kRes = 8;
        for (int i0 = threadIdx.x; i0 < b.nbins; i0 += kRes * kScatThreads) {
            unsigned c[kRes], gp[kRes];
#pragma unroll
            for (int u = 0; u < kRes; ++u) {
                const int i = i0 + u * kScatThreads;
                c[u] = i < b.nbins ? lds_hist[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < kRes; ++u) {
                gp[u] = 0;
                if (c[u]) gp[u] = atomicAdd(&mine[i0 + u * kScatThreads], c[u]);
            }
#pragma unroll
            for (int u = 0; u < kRes; ++u)
                if (c[u]) lds_hist[i0 + u * kScatThreads] = gp[u];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kScatPer; ++k) {
        if (rank[k] == 0xFFFFFFFFu) continue;
        records[lds_hist[rank[k] >> 16] + (rank[k] & 0xFFFFu)] = rec[k];
    }
    if constexpr (Maker::kFixup) {
        // the listed points, one per lane: everything is recomputed from the point itself (its x, y, value and channels come
        // back from the L2), the record overwrites the null record the main loop stored in its slot
        __syncthreads();                                     // (fix_count is final; orders this workgroup's stores to the slot)
        const unsigned nfix = fix_count;
        for (unsigned f = threadIdx.x; f < nfix; f += kScatThreads) {
            const uint2 e = fix_list[f];
            const uint64_t i = base + e.x;
            PointGeom pg;
            const Routed16 r = classify<Maker::kCentre>(g, b, i, x[i], y[i], pg);          // the same verdict as in the main loop
            uint4 rc = make_uint4(kNullCell, 0u, 0u, 0u);
            if (mk.fixup(g, b, r, pg, i, v[i], rc)) records[lds_hist[e.y >> 16] + (e.y & 0xFFFFu)] = rc;
            else fb_list[atomicAdd(fb_count, 1u)] = (unsigned)i;
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------
// Scratch of one binning pass, carved from the engine's arena by the caller (which may append its own needs).
struct Layout {
    size_t o_cnt, o_cursor, o_start, o_nitems, o_fbc, o_items, o_fbl, o_rec, end;
    int max_items;
};

inline Layout layout(size_t base, int max_bins, uint64_t n, unsigned item_records) {
    Layout L;
    size_t off = base;
    auto carve = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
    L.max_items = max_bins + (int)(n / item_records) + 1;
    L.o_cnt = carve((size_t)kVx * max_bins * 4);
    L.o_cursor = carve((size_t)kVx * max_bins * 4);
    L.o_start = carve((size_t)(max_bins + 1) * 4);
    L.o_nitems = carve(4);
    L.o_fbc = carve(4);
    L.o_items = carve((size_t)L.max_items * sizeof(BinItem));
    L.o_fbl = carve((size_t)n * 4);
    L.o_rec = carve((size_t)n * 16);
    L.end = off;
    return L;
}

struct Buffers {
    const uint4* records;
    const unsigned* bin_start;  // [nbins + 1]
    const BinItem* items;
    const unsigned* n_items;
    int max_items;
    unsigned* fb_list;          // points the record cannot hold (indices), painted by the caller's direct kernel
    unsigned* fb_count;         // zeroed by the caller once per scatter (bands append)
};

// tiles per pass: the histograms of both passes live in LDS (4 B per tile)
inline int max_bins(const pcr_hip_engine* e) {
    // PCR_HIP_DEBUG_MAX_BINS lowers every binning pass's limit so that tests reach the banded forms on small grids
    return e->max_bins == kMaxBins ? 16384 : e->max_bins;
}

// What the scatter pass spends its time on (ablation on 50 M points into 14 555 tiles, profiles/r03_b16_scatter.md):
// loads + routing + ranking 0.25 ms, the per-(workgroup, tile) reservations 0.12, the 0.8 GB of records as coalesced
// stores would add 0.09 -- and the fact that every record is its own 16-byte store into a different 128-byte line adds
// 0.39: 50 M store requests, whatever their size.  Runs only get longer with fewer tiles, and the tile kernels want
// their records in LDS: that trade is the path's remaining cost.
// Workgroup size of the scatter pass: 512 -- two workgroups per CU, one's loads and stores behind the other's ranking and
// reservations (measured against one 1024-thread workgroup per CU, 50 M points: 0.69 vs 0.86 ms at 7 313 tiles, 0.62 vs
// 0.91 at 3 249, 0.80 vs 0.89 at 16 384).
template <class Maker>
inline int chunk_of() { return kScatThreads * Maker::kPer; }

// The three passes for the points gd owns (a band: the engine's grid with the owned rows narrowed).  The engine's
// scratch must already hold L.end bytes.
template <class Maker>
int bin(pcr_hip_engine* e, const GridDev& gd, const BinGeom& b, const Maker& mk, const double* x, const double* y,
        const float* v, uint64_t n, unsigned item_records, const Layout& L, Buffers* out) {
    char* s = e->d_scratch;
    auto U = [&](size_t o) { return reinterpret_cast<unsigned*>(s + o); };
    constexpr int chunk = kScatThreads * Maker::kPer;
    static_assert(chunk % 1024 == 0 && Maker::kPer % 4 == 0, "chunk shape");
    if (b.chunk != chunk) return fail(PCR_HIP_INVALID_ARGUMENT, "bin16: BinGeom.chunk must be b16::chunk_of<Maker>()");
    const int blocks = (int)((n + chunk - 1) / chunk);
    const size_t lds = (size_t)b.nbins * 4;
    // the scatter pass of a kFixup Maker lists up to a whole chunk of points beside its histogram
    const size_t lds_scatter = Maker::kFixup ? (size_t)((b.nbins + 3) & ~3) * 4 + (size_t)chunk * sizeof(uint2) : lds;
    PCR_HIP_TRY(hipMemsetAsync(U(L.o_cnt), 0, (size_t)kVx * b.nbins * 4, e->stream));
    {
        ScopedKernelTimer t(e, "k_b16_count");
        auto kernel = &k_b16_count<Maker::kCentre>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        // chunks per counting workgroup: enough that the histogram flush (nbins atomics) is < ~1/4 atomic per point, while
        // the launch still has a few workgroups per CU
        int cb = 1;
        while (cb < 8 && (int64_t)cb * chunk < 4LL * b.nbins && blocks / (cb * 2) >= 3 * e->num_cus) cb *= 2;
        const int cblocks = kVx * (((blocks + kVx - 1) / kVx + cb - 1) / cb);
        hipLaunchKernelGGL(kernel, dim3(cblocks), dim3(kCountThreads), lds, e->stream, gd, b, cb, x, y, n,
                           U(L.o_cnt), U(L.o_fbl), U(L.o_fbc), e->d_touched, e->d_counters);
    }
    {
        ScopedKernelTimer t(e, "k_b16_scan");
        hipLaunchKernelGGL(k_b16_scan, dim3((b.nbins + 1023) / 1024), dim3(1024), 0, e->stream, b.nbins, item_records, U(L.o_cnt), U(L.o_cursor),
                           U(L.o_start), reinterpret_cast<BinItem*>(s + L.o_items), U(L.o_nitems));
    }
    {
        ScopedKernelTimer t(e, "k_b16_scatter");
        auto go = [&](auto kernel) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_scatter);
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kScatThreads), lds_scatter, e->stream, gd, b, mk, x, y, v, n,
                               U(L.o_cursor), reinterpret_cast<uint4*>(s + L.o_rec), U(L.o_fbl), U(L.o_fbc));
        };
        go(&k_b16_scatter<Maker>);
    }
    PCR_HIP_TRY(hipGetLastError());
    out->records = reinterpret_cast<const uint4*>(s + L.o_rec);
    out->bin_start = U(L.o_start);
    out->items = reinterpret_cast<const BinItem*>(s + L.o_items);
    out->n_items = U(L.o_nitems);
    out->max_items = L.max_items;
    out->fb_list = U(L.o_fbl);
    out->fb_count = U(L.o_fbc);
    e->stats_scatter_chunk = chunk;
    return PCR_HIP_OK;
}

}  // namespace b16
}  // namespace pcrhip
