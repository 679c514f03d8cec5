// scatter_binned.hip -- the binned LDS-tile scatter path (Point glyph).
//
// Random per-point atomics into a 134 MB state run at the memory-side atomic rate
// (~9 G atomics/s measured, scatter_direct.hip).  Here the grid is cut into LDS tiles
// ("bins", e.g. 128x128 cells); points are counting-sorted by bin (one histogram pass, one
// scatter pass with LDS-staged, coalesced record writes), then ONE workgroup per bin folds its
// records into an LDS copy of the tile with LDS atomics and merges the tile into the HBM state
// planes with plain coalesced read-modify-writes (a bin is owned by exactly one workgroup per
// launch, so no global atomics at all unless a hot bin had to be split).
//
//   k_bin_count    x,y            -> bin histogram (LDS) -> global bin counts, touched tiles
//   k_bin_scan     counts         -> bin starts, cursors, work items (heavy bins are split)
//   k_bin_scatter  x,y,v          -> records {local cell, value} grouped by bin
//   k_tile_accum   records, state -> state   (LDS atomics + merge pass)
//
// Replaces the reference's sort-by-(tile,cell) + per-tile accumulate
// (src/engine/tile_router_kernels.cu:63-293, src/engine/accumulator_kernels.cu:31-133).
#include "engine.hpp"

using namespace pcrhip;

namespace {

constexpr int kThreads = 1024;          // all four kernels' heavy phases use full-CU workgroups
constexpr int kMaxBins = 4096;
constexpr int kLcellBits = 15;          // up to 32768 cells per LDS tile
constexpr unsigned kLcellMask = (1u << kLcellBits) - 1;
constexpr int kItemRecords = 1 << 17;   // a bin with more records than this is split into several work items

struct BinGeom {
    int tile_w, tile_h;                 // LDS tile, cells
    int bins_x, bins_y, nbins;
    int chunk;                          // points per workgroup in the count / scatter passes
};

struct Record {                         // 8 bytes
    unsigned lcell;
    float value;
};

// LDS tile shape by number of planes: 128 KB of tile per workgroup at most.
inline BinGeom bin_geom(const GridDev& g, uint32_t mask) {
    int planes = __builtin_popcount(mask);
    BinGeom b;
    b.tile_w = 128;
    b.tile_h = planes <= 2 ? 128 : 64;
    b.bins_x = (g.W + b.tile_w - 1) / b.tile_w;
    b.bins_y = (g.st_rows + b.tile_h - 1) / b.tile_h;
    b.nbins = b.bins_x * b.bins_y;
    b.chunk = b.nbins <= 2048 ? 16384 : 8192;
    return b;
}

// ---- shared per-point routing --------------------------------------------------------------
struct Routed {
    bool valid;
    int bin;
    unsigned lcell;
    int row, col;
};

__device__ __forceinline__ Routed route(const GridDev& g, const BinGeom& b, double wx, double wy) {
    Routed r;
    r.valid = world_to_cell(g, wx, wy, r.col, r.row);
    r.valid = r.valid && r.row >= g.own_r0 && r.row < g.own_r1;
    int sr = r.row - g.st_r0;
    int bx = r.col / b.tile_w, by = sr / b.tile_h;
    r.bin = by * b.bins_x + bx;
    r.lcell = (unsigned)((sr - by * b.tile_h) * b.tile_w + (r.col - bx * b.tile_w));
    return r;
}

// ---- pass A: histogram ----------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
k_bin_count(GridDev g, BinGeom b, const double* __restrict__ x, const double* __restrict__ y,
            uint64_t n, unsigned* __restrict__ bin_count, uint32_t* __restrict__ touched,
            unsigned long long* __restrict__ counters) {
    extern __shared__ unsigned lds_hist[];
    for (int i = threadIdx.x; i < b.nbins; i += kThreads) lds_hist[i] = 0;
    __shared__ unsigned any_valid;
    if (threadIdx.x == 0) any_valid = 0;
    __syncthreads();
    const bool one_tile = g.tiles_x * g.tiles_y == 1;
    const uint64_t base = (uint64_t)blockIdx.x * b.chunk;
    unsigned my_valid = 0;
    for (int k = threadIdx.x; k < b.chunk; k += kThreads) {
        uint64_t i = base + k;
        if (i >= n) break;
        Routed r = route(g, b, x[i], y[i]);
        if (r.valid) {
            atomicAdd(&lds_hist[r.bin], 1u);
            ++my_valid;
            if (!one_tile) touch_tile(g, touched, r.row, r.col);
        }
    }
    if (my_valid) atomicAdd(&any_valid, my_valid);
    __syncthreads();
    for (int i = threadIdx.x; i < b.nbins; i += kThreads) {
        unsigned c = lds_hist[i];
        if (c) atomicAdd(&bin_count[i], c);
    }
    if (threadIdx.x == 0 && any_valid) {
        atomicAdd(counters, (unsigned long long)any_valid);
        if (one_tile) touched[0] = 1u;
    }
}

// ---- scan: bin starts + work items ------------------------------------------------------------
// items: {bin, first record, record count, shared flag}; a bin's records are split into items of
// at most kItemRecords so that one hot bin cannot serialize the launch on one CU.
struct Item {
    unsigned bin, first, count, shared;
};

__global__ void __launch_bounds__(kThreads)
k_bin_scan(int nbins, const unsigned* __restrict__ bin_count, unsigned* __restrict__ bin_start,
           unsigned* __restrict__ cursor, Item* __restrict__ items, unsigned* __restrict__ n_items) {
    __shared__ unsigned part[kThreads];
    __shared__ unsigned ipart[kThreads];
    const int per = (nbins + kThreads - 1) / kThreads;
    const int lo = threadIdx.x * per, hi = min(lo + per, nbins);
    unsigned s = 0, it = 0;
    for (int i = lo; i < hi; ++i) {
        unsigned c = bin_count[i];
        s += c;
        it += (c + kItemRecords - 1) / kItemRecords;
    }
    part[threadIdx.x] = s;
    ipart[threadIdx.x] = it;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int off = 1; off < kThreads; off <<= 1) {
        unsigned a = 0, c2 = 0;
        if ((int)threadIdx.x >= off) { a = part[threadIdx.x - off]; c2 = ipart[threadIdx.x - off]; }
        __syncthreads();
        part[threadIdx.x] += a;
        ipart[threadIdx.x] += c2;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - s;        // exclusive prefixes of this thread's span
    unsigned irun = ipart[threadIdx.x] - it;
    for (int i = lo; i < hi; ++i) {
        unsigned c = bin_count[i];
        bin_start[i] = run;
        cursor[i] = run;
        unsigned pieces = (c + kItemRecords - 1) / kItemRecords;
        for (unsigned p = 0; p < pieces; ++p) {
            unsigned first = run + p * kItemRecords;
            unsigned cnt = min((unsigned)kItemRecords, c - p * kItemRecords);
            items[irun + p] = Item{(unsigned)i, first, cnt, pieces > 1 ? 1u : 0u};
        }
        run += c;
        irun += pieces;
    }
    if (threadIdx.x == kThreads - 1) {
        bin_start[nbins] = part[threadIdx.x];
        *n_items = ipart[threadIdx.x];
    }
}

// ---- pass B: scatter records, staged through LDS so that every bin's run is written contiguously
template <int PER_THREAD>
__global__ void __launch_bounds__(kThreads)
k_bin_scatter(GridDev g, BinGeom b, const double* __restrict__ x, const double* __restrict__ y,
              const float* __restrict__ v, uint64_t n, unsigned* __restrict__ cursor,
              Record* __restrict__ records) {
    extern __shared__ unsigned char lds_raw[];
    // layout: stage[chunk] (8 B each) | hist[nbins] | loff[nbins] | gbase[nbins]
    uint2* stage = reinterpret_cast<uint2*>(lds_raw);
    unsigned* hist = reinterpret_cast<unsigned*>(lds_raw + (size_t)b.chunk * sizeof(uint2));
    unsigned* loff = hist + b.nbins;
    unsigned* gbase = loff + b.nbins;
    __shared__ unsigned wave_tot[kThreads / 64];

    for (int i = threadIdx.x; i < b.nbins; i += kThreads) hist[i] = 0;
    __syncthreads();

    const uint64_t base = (uint64_t)blockIdx.x * b.chunk;
    unsigned key[PER_THREAD], rank[PER_THREAD];
    float val[PER_THREAD];
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        uint64_t i = base + (uint64_t)k * kThreads + threadIdx.x;
        key[k] = 0xFFFFFFFFu;
        rank[k] = 0;
        val[k] = 0.0f;
        if (i < n) {
            Routed r = route(g, b, x[i], y[i]);
            if (r.valid) {
                key[k] = ((unsigned)r.bin << kLcellBits) | r.lcell;
                rank[k] = atomicAdd(&hist[r.bin], 1u);         // rank of the point inside (block, bin)
                if (v) val[k] = v[i];
            }
        }
    }
    __syncthreads();

    // block-wide exclusive scan of hist -> loff; reserve global ranges per non-empty bin
    const int per = (b.nbins + kThreads - 1) / kThreads;       // <= 4
    const int lo = threadIdx.x * per, hi = min(lo + per, b.nbins);
    unsigned s = 0;
    for (int i = lo; i < hi; ++i) s += hist[i];
    // wave scan
    unsigned incl = s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += wave_tot[w];
    unsigned run = wave_base + incl - s;
    for (int i = lo; i < hi; ++i) {
        unsigned c = hist[i];
        loff[i] = run;
        if (c) gbase[i] = atomicAdd(&cursor[i], c);
        run += c;
    }
    unsigned total = 0;
    for (int w = 0; w < kThreads / 64; ++w) total += wave_tot[w];
    __syncthreads();

    // stage records grouped by bin
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        if (key[k] != 0xFFFFFFFFu) {
            unsigned bin = key[k] >> kLcellBits;
            stage[loff[bin] + rank[k]] = make_uint2(key[k], __float_as_uint(val[k]));
        }
    }
    __syncthreads();

    // write out: consecutive staged records of a bin go to consecutive global slots
    for (unsigned j = threadIdx.x; j < total; j += kThreads) {
        uint2 rec = stage[j];
        unsigned bin = rec.x >> kLcellBits;
        unsigned dst = gbase[bin] + (j - loff[bin]);
        reinterpret_cast<uint2*>(records)[dst] = make_uint2(rec.x & kLcellMask, rec.y);
    }
}

// ---- pass C: fold a work item's records into an LDS tile, merge the tile into the state planes
template <unsigned MASK>
__global__ void __launch_bounds__(kThreads)
k_tile_accum(GridDev g, BinGeom b, PlanesDev pl, const Record* __restrict__ records,
             const Item* __restrict__ items, const unsigned* __restrict__ n_items) {
    extern __shared__ float lds_tile[];
    constexpr int kPlanes = ((MASK & 1) ? 1 : 0) + ((MASK & 2) ? 1 : 0) + ((MASK & 4) ? 1 : 0) + ((MASK & 8) ? 1 : 0);
    if (blockIdx.x >= *n_items) return;
    const Item it = items[blockIdx.x];
    const int cells = b.tile_w * b.tile_h;
    float* t_sum = lds_tile;
    float* t_wgt = t_sum + ((MASK & 1) ? cells : 0);
    float* t_max = t_wgt + ((MASK & 2) ? cells : 0);
    float* t_min = t_max + ((MASK & 4) ? cells : 0);
    (void)kPlanes;

    for (int i = threadIdx.x; i < cells; i += kThreads) {
        if (MASK & 1) t_sum[i] = 0.0f;
        if (MASK & 2) t_wgt[i] = 0.0f;
        if (MASK & 4) t_max[i] = -FLT_MAX;
        if (MASK & 8) t_min[i] = FLT_MAX;
    }
    __syncthreads();

    const uint2* rec = reinterpret_cast<const uint2*>(records) + it.first;
    for (unsigned j = threadIdx.x; j < it.count; j += kThreads) {
        uint2 r = rec[j];
        float val = __uint_as_float(r.y);
        if (MASK & 1) atomic_add_f32(&t_sum[r.x], val);
        if (MASK & 2) atomic_add_f32(&t_wgt[r.x], 1.0f);
        if (MASK & 4) atomic_max_f32(&t_max[r.x], val);
        if (MASK & 8) atomic_min_f32(&t_min[r.x], val);
    }
    __syncthreads();

    // merge pass: tile -> HBM planes.  Exclusive owner => plain RMW; a split bin => atomics.
    const int bx = it.bin % b.bins_x, by = it.bin / b.bins_x;
    const int c0 = bx * b.tile_w, r0 = by * b.tile_h;                 // r0 relative to the state window
    const int w = min(b.tile_w, g.W - c0), h = min(b.tile_h, g.st_rows - r0);
    for (int i = threadIdx.x; i < b.tile_w * h; i += kThreads) {
        int ly = i / b.tile_w, lx = i - ly * b.tile_w;
        if (lx >= w) continue;
        int64_t cell = (int64_t)(r0 + ly) * g.W + (c0 + lx);
        int li = ly * b.tile_w + lx;
        if (!it.shared) {
            if (MASK & 1) { float a = t_sum[li]; if (a != 0.0f) pl.sum[cell] += a; }
            if (MASK & 2) { float a = t_wgt[li]; if (a != 0.0f) pl.wgt[cell] += a; }
            if (MASK & 4) { float a = t_max[li]; if (a != -FLT_MAX) pl.mx[cell] = fmaxf(pl.mx[cell], a); }
            if (MASK & 8) { float a = t_min[li]; if (a != FLT_MAX) pl.mn[cell] = fminf(pl.mn[cell], a); }
        } else {
            if (MASK & 1) { float a = t_sum[li]; if (a != 0.0f) atomic_add_f32(pl.sum + cell, a); }
            if (MASK & 2) { float a = t_wgt[li]; if (a != 0.0f) atomic_add_f32(pl.wgt + cell, a); }
            if (MASK & 4) { float a = t_max[li]; if (a != -FLT_MAX) atomic_max_f32(pl.mx + cell, a); }
            if (MASK & 8) { float a = t_min[li]; if (a != FLT_MAX) atomic_min_f32(pl.mn + cell, a); }
        }
    }
}

template <unsigned MASK>
void launch_accum(pcr_hip_engine* e, const BinGeom& b, const PlanesDev& pl, const Record* rec,
                  const Item* items, const unsigned* n_items, int max_items) {
    int planes = __builtin_popcount(MASK);
    size_t lds = (size_t)b.tile_w * b.tile_h * sizeof(float) * planes;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_accum<MASK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_tile_accum<MASK>), dim3(max_items), dim3(kThreads), lds, e->stream, e->gd, b, pl,
                       rec, items, n_items);
}

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace

namespace pcrhip {

bool binned_point_supported(const pcr_hip_engine* e, uint32_t mask) {
    if (mask == 0 || (mask & ~15u)) return false;
    BinGeom b = bin_geom(e->gd, mask);
    if (b.nbins > kMaxBins) return false;
    // not worth the fixed cost of sweeping every tile for a handful of points
    uint64_t cells = (uint64_t)e->gd.W * e->gd.st_rows;
    if (e->forced_path != 2 && e->stats.points_in * 16 < cells) return false;
    return e->stats.points_in < (1ull << 32) - (1ull << 20);
}

int binned_point(pcr_hip_engine* e, uint32_t mask, const PlanesDev& pl,
                 const double* x, const double* y, const float* v, uint64_t n) {
    const BinGeom b = bin_geom(e->gd, mask);
    const int blocks = (int)((n + b.chunk - 1) / b.chunk);
    const int max_items = b.nbins + (int)(n / kItemRecords) + 1;

    // scratch carve-up
    size_t off = 0;
    const size_t o_count = off;  off += align256((size_t)b.nbins * 4);
    const size_t o_start = off;  off += align256((size_t)(b.nbins + 1) * 4);
    const size_t o_cursor = off; off += align256((size_t)b.nbins * 4);
    const size_t o_nitems = off; off += 256;
    const size_t o_items = off;  off += align256((size_t)max_items * sizeof(Item));
    const size_t o_rec = off;    off += align256((size_t)n * sizeof(Record));
    int rc = ensure_scratch(e, off);
    if (rc) return rc;
    char* s = e->d_scratch;
    unsigned* d_count = reinterpret_cast<unsigned*>(s + o_count);
    unsigned* d_start = reinterpret_cast<unsigned*>(s + o_start);
    unsigned* d_cursor = reinterpret_cast<unsigned*>(s + o_cursor);
    unsigned* d_nitems = reinterpret_cast<unsigned*>(s + o_nitems);
    Item* d_items = reinterpret_cast<Item*>(s + o_items);
    Record* d_rec = reinterpret_cast<Record*>(s + o_rec);

    PCR_HIP_TRY(hipMemsetAsync(d_count, 0, (size_t)b.nbins * 4, e->stream));
    {
        ScopedKernelTimer t(e, "k_bin_count");
        hipLaunchKernelGGL(k_bin_count, dim3(blocks), dim3(kThreads), (size_t)b.nbins * 4, e->stream,
                           e->gd, b, x, y, n, d_count, e->d_touched, e->d_counters);
    }
    {
        ScopedKernelTimer t(e, "k_bin_scan");
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(kThreads), 0, e->stream, b.nbins, d_count, d_start,
                           d_cursor, d_items, d_nitems);
    }
    {
        ScopedKernelTimer t(e, "k_bin_scatter");
        size_t lds = (size_t)b.chunk * sizeof(uint2) + (size_t)b.nbins * 4 * 3;
        if (b.chunk == 16384) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_scatter<16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k_bin_scatter<16>), dim3(blocks), dim3(kThreads), lds, e->stream, e->gd, b,
                               x, y, v, n, d_cursor, d_rec);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_scatter<8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k_bin_scatter<8>), dim3(blocks), dim3(kThreads), lds, e->stream, e->gd, b,
                               x, y, v, n, d_cursor, d_rec);
        }
    }
    {
        ScopedKernelTimer t(e, "k_tile_accum");
        switch (mask) {
#define PCR_ACC(M) case M: launch_accum<M>(e, b, pl, d_rec, d_items, d_nitems, max_items); break;
            PCR_ACC(1) PCR_ACC(2) PCR_ACC(3) PCR_ACC(4) PCR_ACC(5) PCR_ACC(6) PCR_ACC(7) PCR_ACC(8)
            PCR_ACC(9) PCR_ACC(10) PCR_ACC(11) PCR_ACC(12) PCR_ACC(13) PCR_ACC(14) PCR_ACC(15)
#undef PCR_ACC
            default: return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_point: empty plane mask");
        }
    }
    PCR_HIP_TRY(hipGetLastError());
    e->stats.path = 1;
    e->stats.lds_tile_w = b.tile_w;
    e->stats.lds_tile_h = b.tile_h;
    e->stats.lds_apron = 0;
    e->stats.num_bins = b.nbins;
    return PCR_HIP_OK;
}

bool binned_glyph_supported(const pcr_hip_engine*, const GlyphDev&, uint32_t) { return false; }
int binned_glyph(pcr_hip_engine*, const GlyphDev&, uint32_t, const PlanesDev&, const double*, const double*,
                 const float*, uint64_t) {
    return fail(PCR_HIP_NOT_IMPLEMENTED, "binned glyph path not built");
}

}  // namespace pcrhip
