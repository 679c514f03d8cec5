// scatter_binned.hip -- the binned LDS-tile scatter path: binning passes + Point-glyph tiles.
//
// Random per-point atomics into a 134 MB state run at the memory-side atomic rate
// (~9 G atomics/s measured, scatter_direct.hip).  Here the grid is cut into LDS tiles
// ("bins"); points are counting-sorted by bin (one histogram pass that also writes a 4-byte
// routing key per point, one scatter pass with LDS-staged, coalesced record writes), then ONE
// workgroup per bin folds its records into an LDS copy of the tile with LDS atomics and merges
// the tile into the HBM state planes with plain coalesced read-modify-writes (a bin is owned by
// exactly one workgroup per launch, so no global atomics unless a hot bin had to be split).
//
//   k_bin_count    x,y            -> routing keys, bin histogram (LDS) -> global bin counts, touched tiles
//   k_bin_scan     counts         -> bin starts, cursors, work items (heavy bins are split)
//   k_bin_scatter  keys,v         -> records {local cell, value | point index} grouped by bin
//   k_tile_accum   records, state -> state   (LDS atomics + merge pass)       [Point glyph]
//   (glyph tiles: scatter_binned_glyph.hip)
//
// Replaces the reference's sort-by-(tile,cell) + per-tile accumulate
// (src/engine/tile_router_kernels.cu:63-293, src/engine/accumulator_kernels.cu:31-133).
#include "engine.hpp"

#include <cstdlib>

using namespace pcrhip;

namespace {

constexpr int kThreads = 1024;          // full-CU workgroups
constexpr unsigned kLcellMask = (1u << kLcellBits) - 1;
constexpr unsigned kPointItemRecords = 1u << 17;   // a bin with more records is split into several work items

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
    int sr = r.row - g.st_r0 - b.row0;                 // valid points lie inside the band: sr >= 0
    int bx = fast_div(r.col, b.tile_w), by = fast_div(sr, b.tile_h);
    r.bin = by * b.bins_x + bx;
    r.lcell = (unsigned)((sr - by * b.tile_h) * b.tile_w + (r.col - bx * b.tile_w));
    return r;
}

// ---- pass A: routing keys + histogram -------------------------------------------------------
// 512-thread workgroups: the kernel needs ~106 SGPRs (7 waves per SIMD), so 1024-thread groups ran one per CU;
// three 512-thread groups per CU keep half as many loads again in flight.
constexpr int kCountThreads = 512;

// Blocks [0, full_blocks) take b.chunk points each, the blocks after them 4096 points each (the ragged end): the SAME
// block -> points mapping as k_bin_scatter's, because with nvx = 8 the counts are kept per virtual XCD (blockIdx % 8) and a
// point must be counted under the virtual XCD that will store its record (see bin_points).  nvx = 1: one count per bin.
// ONE_TILE: the grid is one reference tile (C2, C4: the whole touched-tile bookkeeping -- its LDS flags, two more divisors and
// their reciprocals -- is compiled out; the workgroup's first thread flags the tile).
template <bool MULTI, bool ONE_TILE>
__global__ void __launch_bounds__(kCountThreads)
k_bin_count(GridDev g_uniform, BinGeom b, unsigned full_blocks, int split, int nvx, const double* __restrict__ x, const double* __restrict__ y,
            uint64_t n, unsigned* __restrict__ keys, unsigned* __restrict__ bin_count,
            uint32_t* __restrict__ touched, unsigned long long* __restrict__ counters) {
    // (common.hpp: the scalar registers do not hold all of it; the one-tile variant needs the doubles only)
    const GridDev g = vector_resident<(ONE_TILE ? (PCR_VRES_POINT > 0 ? 1 : 0) : PCR_VRES_POINT)>(g_uniform);
    extern __shared__ unsigned lds_hist[];
    for (int i = threadIdx.x; i < b.nbins; i += kCountThreads) lds_hist[i] = 0;
    __shared__ unsigned any_valid;
    __shared__ unsigned lds_touch[ONE_TILE ? 1 : kTouchLdsTiles];
    TouchLds tl;
    if (!ONE_TILE) tl.begin(g, lds_touch, kCountThreads);
    if (threadIdx.x == 0) any_valid = 0;
    __syncthreads();
    constexpr bool one_tile = ONE_TILE;
    // Block -> points.  MULTI = false: a scatter block's points are counted by `split` workgroups (all under the scatter
    // block's virtual XCD) -- the count pass wants more, shorter workgroups than the scatter pass has chunks.  MULTI = true
    // (`split` then holds cb): ONE workgroup counts cb scatter blocks of one virtual XCD (blocks vx, vx + 8, ...) -- with
    // ~100 bins, the first level of the two-level sort, every workgroup flushes onto the same few hundred counters, and
    // 500 M points on a 16384 x 8192 window made 40 690 such workgroups: k_bin_count 2.44 -> 1.90-1.99 ms with eight blocks
    // each (count_blocks()).  (A separate instantiation: as one kernel with a loop of one, the C2 count pass lost 9 %.)
    // The ragged end's 4096-point blocks come last, one workgroup each.
    const unsigned nsplit = MULTI ? (unsigned)nvx * (((full_blocks + nvx - 1) / nvx + split - 1) / split) : full_blocks * (unsigned)split;
    const bool tail = blockIdx.x >= nsplit;
    const unsigned sblock = tail ? full_blocks + (blockIdx.x - nsplit)                               // the scatter pass's block
                          : MULTI ? (blockIdx.x / (unsigned)nvx) * (unsigned)split * (unsigned)nvx + (blockIdx.x & (unsigned)(nvx - 1))
                                  : blockIdx.x / (unsigned)split;
    const int len = tail ? 4096 : MULTI ? b.chunk : b.chunk / split;
    uint64_t base = tail ? (uint64_t)full_blocks * b.chunk + (uint64_t)(blockIdx.x - nsplit) * 4096
                         : MULTI ? (uint64_t)sblock * b.chunk
                                 : (uint64_t)sblock * b.chunk + (uint64_t)(blockIdx.x % (unsigned)split) * len;
    unsigned my_valid = 0;
    // The routing is done once: pass B reads the 4-byte key written here instead of x, y (16 B).
    auto handle = [&](uint64_t i, double wx, double wy) -> unsigned {
        Routed r = route(g, b, wx, wy);
        if (r.valid && point_kept(g, i)) {
            atomicAdd(&lds_hist[r.bin >> b.sup_shift], 1u);
            ++my_valid;
            if (!one_tile) tl.touch(g, touched, r.row, r.col);
            return ((unsigned)r.bin << kLcellBits) | r.lcell;
        }
        return 0xFFFFFFFFu;
    };
    // (MULTI: the last group of workgroups may start beyond the last full block -- full_blocks need not be a multiple of
    // nvx -- and such a workgroup has nothing to count: its `base` would be the start of the ragged end, which the
    // 4096-point blocks count.  It still reaches the barriers and the (empty) flush below.)
    const int nranges = MULTI && !tail ? (sblock >= full_blocks ? 0 : split) : 1;
    for (int t = 0; t < nranges; ++t) {
    if (MULTI && t > 0) {
        if (sblock + (unsigned)t * (unsigned)nvx >= full_blocks) break;
        base += (uint64_t)nvx * b.chunk;
    }
    const bool full = base + (uint64_t)len <= n &&
                      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (full) {
        // 16-byte loads (two points per lane), four of them in flight per array before any math (len is a multiple of 1024)
        const double2* x2 = reinterpret_cast<const double2*>(x + base);
        const double2* y2 = reinterpret_cast<const double2*>(y + base);
        uint2* k2 = reinterpret_cast<uint2*>(keys + base);
        const int pairs = len >> 1;
        for (int p0 = threadIdx.x; p0 < pairs; p0 += 4 * kCountThreads) {
            double2 xs[4], ys[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p0 + u * kCountThreads >= pairs) break;               // (uniform over the workgroup)
                xs[u] = stream_load(x2 + p0 + u * kCountThreads);
                ys[u] = stream_load(y2 + p0 + u * kCountThreads);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p0 + u * kCountThreads >= pairs) break;
                const uint64_t i = base + 2ull * (p0 + u * kCountThreads);
                unsigned ka = handle(i, xs[u].x, ys[u].x);
                unsigned kb = handle(i + 1, xs[u].y, ys[u].y);
                // (non-temporal: 200 MB of keys, read once by pass B; C2 step -0.5 %)
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                __builtin_nontemporal_store(u2v{ka, kb}, reinterpret_cast<u2v*>(k2 + p0 + u * kCountThreads));
            }
        }
    } else {
        for (int k = threadIdx.x; k < len; k += kCountThreads) {
            uint64_t i = base + k;
            if (i >= n) break;
            keys[i] = handle(i, x[i], y[i]);
        }
    }
    }
    if (my_valid) atomicAdd(&any_valid, my_valid);
    __syncthreads();
    unsigned* mine = bin_count + (size_t)(sblock & (unsigned)(nvx - 1)) * b.nbins;
    for (int i = threadIdx.x; i < b.nbins; i += kCountThreads) {
        unsigned c = lds_hist[i];
        if (c) atomicAdd(&mine[i], c);
    }
    if (!ONE_TILE) tl.flush(g, touched, kCountThreads);
    if (threadIdx.x == 0 && any_valid) {
        atomicAdd(counters, (unsigned long long)any_valid);
        if (one_tile) touched[0] = 1u;
    }
}

template <bool MULTI, class... Args>
void launch_bin_count(bool one_tile, dim3 grid, size_t lds, hipStream_t stream, Args... args) {
    if (one_tile) hipLaunchKernelGGL((k_bin_count<MULTI, true>), grid, dim3(kCountThreads), lds, stream, args...);
    else hipLaunchKernelGGL((k_bin_count<MULTI, false>), grid, dim3(kCountThreads), lds, stream, args...);
}

// Points per workgroup of the count pass.  Every block flushes its LDS histogram with up to nbins global atomics, so many
// bins want long chunks -- but long chunks mean few blocks in flight.  Measured in round 2, k_bin_count ms:
//   chunk          8192    16384   32768   65536   131072
//   1376 bins (C2, 50 M points)        0.228           0.267
//   2816 bins (16384 x 2048, 125 M)  0.672   0.648   0.630   0.647   0.697
//   5504 bins (16384 x 4096, 250 M)  1.260   1.193   1.154   1.178   1.231
// The single-level path now counts in the scatter pass's own blocks (scatter_shape: 12 288 or 16 384 points), because
// the counts are kept per virtual XCD and both passes must see a point in the same block (bin_points).

// Workgroups of the count pass per scatter block.  The scatter pass wants long chunks (its (block, bin) runs), the count pass
// short ones: with round 2's 28 672-point chunks its 1 744 workgroups of 512 threads were 2.3 rounds of the 768 a launch keeps
// resident, and the third, quarter-full round ran as long as a full one.  Four workgroups per scatter block (7 168 points
// each, counted under the scatter block's virtual XCD), A/B in one call on C2: k_bin_count 0.202 / 0.207 -> 0.182 / 0.186 ms,
// the step 0.606 / 0.612 -> 0.591 / 0.592 (two: 0.194 / 0.180; seven: 0.208 / 0.213 -- every workgroup flushes its histogram
// with up to nbins global atomics, so the split stops where that exceeds ~1/5 atomic per point).  With today's chunks
// (scatter_shape: 12 288 points at C2's 1 376 bins) the rule leaves one workgroup per block: 4 069 of them, 5.3 rounds.
// Scatter blocks per count workgroup (k_bin_count<true>): with few bins every workgroup's flush lands on the same few hundred
// counters, and what matters is how MANY workgroups flush -- as many blocks per workgroup as still leave ~4 rounds of
// workgroups (768 resident).  1: the split rule below applies instead.
inline int count_blocks(int nbins, int blocks, int num_cus) {
    if (nbins > 512) return 1;
    int cb = 1;
    while (cb < 16 && blocks / (cb * 2) >= 12 * num_cus) cb *= 2;
    return cb;
}

inline int count_split(int chunk, int nbins) {
    int s = 4;
    while (s > 1 && (chunk % s != 0 || (chunk / s) % 1024 != 0 || chunk / s < 5 * nbins)) s >>= 1;
    return s;
}

// ---- scan: bin starts + work items ------------------------------------------------------------
// A bin's records are split into items of at most item_records so that one hot bin cannot
// serialize the launch on one CU.
// nvx > 1: counts and cursors are laid out [virtual XCD][bin]; a bin's records are [vx 0 | vx 1 | ...], contiguous.
// every_bin: a bin without records still gets one (empty) item, so that the tile pass visits every cell (state
// initialisation inside the scatter).  n_items[1] = 1 when some bin was split into several items.
__global__ void __launch_bounds__(kThreads)
k_bin_scan(int nbins, int nvx, unsigned item_records, const unsigned* __restrict__ bin_count,
           unsigned* __restrict__ cursor, BinItem* __restrict__ items, unsigned* __restrict__ n_items, int every_bin) {
    __shared__ unsigned part[kThreads];
    __shared__ unsigned ipart[kThreads];
    const int per = (nbins + kThreads - 1) / kThreads;
    const int lo = threadIdx.x * per, hi = min(lo + per, nbins);
    auto total = [&](int i) {
        unsigned c = 0;
        for (int v = 0; v < nvx; ++v) c += bin_count[(size_t)v * nbins + i];
        return c;
    };
    unsigned s = 0, it = 0;
    int split = 0;
    for (int i = lo; i < hi; ++i) {
        unsigned c = total(i);
        s += c;
        const unsigned pieces = (c + item_records - 1) / item_records;
        it += (every_bin && pieces == 0) ? 1u : pieces;
        split |= pieces > 1;
    }
    split = __syncthreads_or(split);
    part[threadIdx.x] = s;
    ipart[threadIdx.x] = it;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {             // Hillis-Steele inclusive scan
        unsigned a = 0, c2 = 0;
        if ((int)threadIdx.x >= off) { a = part[threadIdx.x - off]; c2 = ipart[threadIdx.x - off]; }
        __syncthreads();
        part[threadIdx.x] += a;
        ipart[threadIdx.x] += c2;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - s;                       // exclusive prefixes of this thread's span
    unsigned irun = ipart[threadIdx.x] - it;
    for (int i = lo; i < hi; ++i) {
        unsigned c = 0;
        for (int v = 0; v < nvx; ++v) {
            cursor[(size_t)v * nbins + i] = run + c;
            c += bin_count[(size_t)v * nbins + i];
        }
        unsigned pieces = (c + item_records - 1) / item_records;
        for (unsigned p = 0; p < pieces; ++p) {
            unsigned first = run + p * item_records;
            unsigned cnt = min(item_records, c - p * item_records);
            items[irun + p] = BinItem{(unsigned)i, first, cnt, pieces > 1 ? 1u : 0u};
        }
        if (every_bin && pieces == 0) { items[irun] = BinItem{(unsigned)i, run, 0u, 0u}; pieces = 1; }
        run += c;
        irun += pieces;
    }
    if (threadIdx.x == kThreads - 1) { n_items[0] = ipart[threadIdx.x]; n_items[1] = split ? 1u : 0u; }
}

// ---- pass B: scatter records, staged through LDS so that every bin's run is written contiguously
//
// One workgroup = 512 threads x PER_THREAD points = one chunk (16384 or 8192 points).  The chunk is ranked by
// bin with LDS atomics, bins are scanned, every non-empty (block, bin) run reserves its place in the bin's
// global range with ONE atomic, and the records leave through an LDS staging WINDOW of 8192 records
// (64 KB): round r stages the records whose position inside the block's sorted order falls in
// [r * window, (r + 1) * window) and writes them out.  Positions, not bins, define the rounds, so a skewed chunk
// costs nothing extra.  Round 1 staged the whole chunk (128 KB): one workgroup per CU, and every phase of it
// (load / rank / reserve / stage / write) ran with nothing else to overlap -- 60 % of the wave-cycles waited
// (profiles/r01_c2_sq_counters.md).  With the window two workgroups share a CU (2 x (64 KB + 8 B per bin)),
// runs stay as long as before (they depend on the chunk, not on the window), and one workgroup's loads and stores
// overlap the other's LDS phases.
// VEC: every block of the launch is a full chunk and keys/v are 16-byte aligned (16-byte loads,
// four consecutive points per lane); the ragged last chunk is a second, scalar launch.
// INDEX: record.y = index of the point instead of its value (Gaussian tiles: LDS-atomic bound, the gather is free).
// Shape = THREADS x PER points per workgroup, WINDOW records staged per round.  Measured on MI355X (50 M points;
// tools/tune_scatter.sh, profiles/r02_tune_scatter.md): what pays is the LENGTH OF THE RUNS, i.e. the chunk --
// every (block, bin) run is a partial-line write, and the 28672-point chunk that the 64 KB window makes possible
// (1024 x 28, the most that stays under 128 VGPRs) beat the 16384-point chunk staged whole by 5 % at 1376 bins [round 2's
// measurement; round 4's, on today's kernels, is at scatter_shape below and says otherwise]
// (C2), 27 % at 2816 bins (a C5 shard) and 35 % at 4096 bins (Gaussian index records).  Two 512-thread workgroups
// per CU (same chunk, window 8192 or 4096) did NOT help: the load and write phases already run at the CU's
// fair share of HBM, what is left is the sub-line write pattern itself.
template <int THREADS, int PER_THREAD, int WINDOW, bool VEC, bool INDEX>
__device__ __forceinline__ void
bin_scatter_chunk(unsigned char* lds_raw, const BinGeom& b, uint64_t base, const unsigned* __restrict__ keys,
                  const float* __restrict__ v, uint64_t n, unsigned* __restrict__ cursor, uint2* __restrict__ records) {
    // layout: stage[window] (8 B each) | hist[nbins] | loff[nbins]; after the reservation hist[bin] holds
    // (global start of the block's run) - loff[bin], so that the record at sorted position j goes to hist[bin] + j
    uint2* stage = reinterpret_cast<uint2*>(lds_raw);
    unsigned* hist = reinterpret_cast<unsigned*>(lds_raw + (size_t)WINDOW * sizeof(uint2));
    unsigned* loff = hist + b.nbins;
    constexpr int kWaves = THREADS / 64;
    __shared__ unsigned wave_tot[kWaves];

    for (int i = threadIdx.x; i < b.nbins; i += THREADS) hist[i] = 0;
    __syncthreads();
    const int gshift = kLcellBits + b.sup_shift;          // key -> bin of this pass (a tile, or a group of tiles)

    unsigned key[PER_THREAD], pos[PER_THREAD], val[PER_THREAD];
    // keys first: the values are only needed when the records are staged, so their loads are issued after the
    // scan (below) and complete behind the reservation atomics -- and the registers they need are not live while the
    // keys are ranked
    if (VEC) {
        const uint4* k4 = reinterpret_cast<const uint4*>(keys + base);
#pragma unroll
        for (int q = 0; q < PER_THREAD / 4; ++q) {
            uint4 kk = stream_load(k4 + q * THREADS + threadIdx.x);
            key[4 * q + 0] = kk.x; key[4 * q + 1] = kk.y; key[4 * q + 2] = kk.z; key[4 * q + 3] = kk.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) {
            uint64_t i = base + (uint64_t)k * THREADS + threadIdx.x;
            key[k] = i < n ? keys[i] : 0xFFFFFFFFu;
        }
    }
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        pos[k] = 0;
        if (key[k] != 0xFFFFFFFFu) pos[k] = atomicAdd(&hist[key[k] >> gshift], 1u);       // rank inside (block, bin)
    }
    __syncthreads();

    // block-wide exclusive scan of the bin counts -> loff (thread t owns the consecutive bins [t*per, (t+1)*per))
    const int per = (b.nbins + THREADS - 1) / THREADS;
    const int lo = threadIdx.x * per, hi = min(lo + per, b.nbins);
    unsigned s = 0;
    for (int i = lo; i < hi; ++i) s += hist[i];
    unsigned incl = s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned run = incl - s;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
    for (int i = lo; i < hi; ++i) {
        loff[i] = run;
        run += hist[i];
    }
    unsigned total = 0;
    for (int w = 0; w < kWaves; ++w) total += wave_tot[w];
    __syncthreads();

    if (VEC) {
        const uint4* v4 = reinterpret_cast<const uint4*>(v + base);
#pragma unroll
        for (int q = 0; q < PER_THREAD / 4; ++q) {
            const unsigned p = q * THREADS + threadIdx.x;
            if (INDEX) {
                unsigned i0 = (unsigned)base + 4u * p;
                val[4 * q + 0] = i0; val[4 * q + 1] = i0 + 1; val[4 * q + 2] = i0 + 2; val[4 * q + 3] = i0 + 3;
            } else {
                uint4 vv = v ? stream_load(v4 + p) : make_uint4(0u, 0u, 0u, 0u);
                val[4 * q + 0] = vv.x; val[4 * q + 1] = vv.y; val[4 * q + 2] = vv.z; val[4 * q + 3] = vv.w;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) {
            uint64_t i = base + (uint64_t)k * THREADS + threadIdx.x;
            val[k] = 0u;
            if (i < n) {
                if (INDEX) val[k] = (unsigned)i;
                else if (v) val[k] = __float_as_uint(v[i]);
            }
        }
    }

    // every non-empty (block, bin) run reserves its place in the bin's global range: bins are dealt to lanes
    // INTERLEAVED (a wave's atomics hit 64 consecutive words) and a lane's atomics are issued back to back --
    // with consecutive ownership they were strided over the cursor array and each waited for the one before
    {
        constexpr int kRes = 4;
        for (int i0 = threadIdx.x; i0 < b.nbins; i0 += kRes * THREADS) {
            unsigned c[kRes], gpos[kRes];
#pragma unroll
            for (int u = 0; u < kRes; ++u) {
                const int i = i0 + u * THREADS;
                c[u] = i < b.nbins ? hist[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < kRes; ++u) {
                gpos[u] = 0;
                if (c[u]) gpos[u] = atomicAdd(&cursor[i0 + u * THREADS], c[u]);
            }
#pragma unroll
            for (int u = 0; u < kRes; ++u)
                if (c[u]) hist[i0 + u * THREADS] = gpos[u] - loff[i0 + u * THREADS];
        }
    }
    // rank inside the bin -> position inside the block's sorted order (loff is final since the last barrier)
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k)
        if (key[k] != 0xFFFFFFFFu) pos[k] += loff[key[k] >> gshift];
    // (the barrier that orders the reservations before the write-out is the one after the first staging round)

    for (unsigned w0 = 0; w0 < total; w0 += WINDOW) {
        // stage the records of this window, grouped by bin
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) {
            const unsigned rel = pos[k] - w0;                  // wraps for positions before the window
            if (key[k] != 0xFFFFFFFFu && rel < (unsigned)WINDOW) stage[rel] = make_uint2(key[k], val[k]);
        }
        __syncthreads();
        // write out: consecutive staged records of a bin go to consecutive global slots
        const unsigned cnt = min((unsigned)WINDOW, total - w0);
        for (unsigned j = threadIdx.x; j < cnt; j += THREADS) {
            uint2 rec = stage[j];
            unsigned bin = rec.x >> gshift;
            unsigned dst = hist[bin] + w0 + j;
            records[dst] = make_uint2(b.sup_shift ? rec.x : rec.x & kLcellMask, rec.y);   // first of two levels: keep the tile
        }
        __syncthreads();
    }
}

// Blocks [0, full_blocks) take a full chunk each (16-byte loads); the blocks after them share the ragged end of the
// cloud in 4096-point chunks through the scalar-load body: one launch (a second, tiny launch was ~10 us of latency).
template <int THREADS, int PER_THREAD, int WINDOW, bool INDEX>
__global__ void __launch_bounds__(THREADS, 4)          // <= 128 VGPRs
k_bin_scatter(BinGeom b, unsigned full_blocks, int nvx, const unsigned* __restrict__ keys, const float* __restrict__ v,
              uint64_t n, unsigned* __restrict__ cursor, uint2* __restrict__ records) {
    extern __shared__ unsigned char lds_dyn[];
    cursor += (size_t)(blockIdx.x & (unsigned)(nvx - 1)) * b.nbins;      // this workgroup's virtual XCD (bin_points)
    if (blockIdx.x < full_blocks) {
        bin_scatter_chunk<THREADS, PER_THREAD, WINDOW, true, INDEX>(lds_dyn, b, (uint64_t)blockIdx.x * (THREADS * PER_THREAD), keys, v, n,
                                                                    cursor, records);
    } else {
        const uint64_t base = (uint64_t)full_blocks * (THREADS * PER_THREAD) + (uint64_t)(blockIdx.x - full_blocks) * 4096;
        bin_scatter_chunk<THREADS, 4096 / THREADS, WINDOW, false, INDEX>(lds_dyn, b, base, keys, v, n, cursor, records);
    }
}

// Shape of the scatter pass: THREADS x PER points per workgroup (the chunk), WINDOW records staged per round.  Round 2
// settled on 1024 x 28 with a 64 KB window (one workgroup per CU at the VGPR cap); round 4 measured the shapes again on the
// kernels as they are now (A/B in one call, tools/ab_libs.sh; dominant-kernel ms of the timed steps / C2 step ms):
//     1024 x 28, 8192   0.222-0.226 / 0.587-0.594        512 x 24, 8192    0.181-0.190 / 0.545-0.553   <- two workgroups per CU
//     1024 x 24, 8192   0.189-0.194 / 0.564-0.567        512 x 20, 8192    0.181-0.186 / 0.551-0.553
//     1024 x 16, 8192   0.193-0.199 / 0.555-0.568        512 x 16, 8192    0.187-0.188 / 0.558-0.562
//     1024 x 16, 16384  0.190-0.194 / 0.557-0.558        512 x 32, 8192    0.220-0.223 / 0.592-0.602
//     1024 x 12, 8192   0.235-0.239 / 0.604-0.611        512 x 32, 4096    0.265-0.270 / 0.632-0.634
// and on a C5 shard (16384 x 2048, 2 816 bins, 125 M points; step ms): 1024 x 28: 1.495, 512 x 24: 1.553 (runs of four
// records), 1024 x 16 with a 16384-record window: 1.428.  So: few bins -> two 512-thread workgroups per CU, whose phases
// overlap; more bins -> the longer chunk of one 1024-thread workgroup, staged in ONE round while the LDS has the room.
struct ScatterShape {
    int threads, per, window;
    int chunk() const { return threads * per; }
};
inline ScatterShape scatter_shape(int nbins) {
    if (nbins <= 2040) return {512, 24, 8192};               // (64 KB + 8 B per bin + the static words, twice, within 160 KB)
    if (nbins <= 4032) return {1024, 16, 16384};            // 128 KB + 8 B per bin (+ the static words) = the whole LDS
    return {1024, 24, 8192};                                // (a quarter of C5, 5 504 bins: step 2.80 -> 2.72-2.76 ms against 1024 x 16; x 28: 2.86-2.95)
}
constexpr int kVirtualXcds = 8;                    // record sub-ranges per bin (bin_points)
// full chunks of the scatter pass (the rest of the cloud goes in 4096-point blocks)
inline int scatter_full_blocks(const ScatterShape& sh, const float* v, uint64_t n, bool index) {
    const bool aligned = index || (reinterpret_cast<uintptr_t>(v) & 15) == 0;     // the keys are 256-B aligned
    return aligned ? (int)(n / (uint64_t)sh.chunk()) : 0;
}

template <bool INDEX>
void launch_bin_scatter(pcr_hip_engine* e, const BinGeom& b, int nvx, const unsigned* d_keys, const float* v, uint64_t n,
                        unsigned* d_cursor, uint2* d_rec) {
    const ScatterShape sh = scatter_shape(b.nbins);
    const uint64_t chunk = (uint64_t)sh.chunk();
    const int full_blocks = scatter_full_blocks(sh, v, n, INDEX);
    const size_t lds = (size_t)sh.window * sizeof(uint2) + (size_t)b.nbins * 4 * 2;
    const uint64_t done = (uint64_t)full_blocks * chunk;
    const unsigned tail_blocks = (unsigned)((n - done + 4095) / 4096);            // the chunk is a multiple of 4096
    ScopedKernelTimer t(e, "k_bin_scatter");
    auto go = [&](auto kernel, int threads) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3((unsigned)full_blocks + tail_blocks), dim3(threads), lds, e->stream, b, (unsigned)full_blocks, nvx,
                           d_keys, v, n, d_cursor, d_rec);
    };
    if (sh.threads == 512) go(&k_bin_scatter<512, 24, 8192, INDEX>, 512);
    else if (sh.window == 16384) go(&k_bin_scatter<1024, 16, 16384, INDEX>, 1024);
    else go(&k_bin_scatter<1024, 24, 8192, INDEX>, 1024);
    e->stats_scatter_chunk = (int)chunk;
}

// ---- second level of the two-level sort (grids with more tiles than one pass can count in LDS) ----
// After the first level the records {tile << 15 | local cell, value} are grouped by GROUP of 2^sup_shift
// consecutive tiles; a work item is at most 16384 records of one group.  k_sub_count histograms an item by
// tile, k_bin_scan turns the tile counts into starts, k_sub_scatter sorts the item by tile through LDS and
// writes {local cell, value} runs -- the same three steps as the first level, on 8-byte records instead of
// x, y.  Both levels write long runs (16384 / groups, 16384 / tiles-per-group records), where the row-band
// sweep scattered single records.
constexpr int kSubPer = 16;                       // records per thread: items of 16384

__global__ void __launch_bounds__(kThreads)
k_sub_count(int sup_shift, const uint2* __restrict__ rec, const BinItem* __restrict__ items,
            const unsigned* __restrict__ n_items, unsigned* __restrict__ tile_count) {
    extern __shared__ unsigned lds_u32[];
    if (blockIdx.x >= *n_items) return;
    const BinItem it = items[blockIdx.x];
    const int tps = 1 << sup_shift;
    for (int i = threadIdx.x; i < tps; i += kThreads) lds_u32[i] = 0;
    __syncthreads();
    const unsigned tile0 = it.bin << sup_shift;
    const uint2* r = rec + it.first;
    for (unsigned j = threadIdx.x; j < it.count; j += kThreads) atomicAdd(&lds_u32[(r[j].x >> kLcellBits) - tile0], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < tps; i += kThreads) {
        const unsigned c = lds_u32[i];
        if (c) atomicAdd(&tile_count[tile0 + i], c);
    }
}

__global__ void __launch_bounds__(kThreads)
k_sub_scatter(int sup_shift, const uint2* __restrict__ rec, const BinItem* __restrict__ items,
              const unsigned* __restrict__ n_items, unsigned* __restrict__ cursor, uint2* __restrict__ out) {
    extern __shared__ unsigned char lds_raw[];
    if (blockIdx.x >= *n_items) return;
    const BinItem it = items[blockIdx.x];
    const int tps = 1 << sup_shift;
    // layout: stage[16384] (8 B each) | hist[tps] | loff[tps] | gbase[tps]
    uint2* stage = reinterpret_cast<uint2*>(lds_raw);
    unsigned* hist = reinterpret_cast<unsigned*>(lds_raw + (size_t)kSubPer * kThreads * sizeof(uint2));
    unsigned* loff = hist + tps;
    unsigned* gbase = loff + tps;
    __shared__ unsigned wave_tot[kThreads / 64];
    for (int i = threadIdx.x; i < tps; i += kThreads) hist[i] = 0;
    __syncthreads();
    const unsigned tile0 = it.bin << sup_shift;
    const uint2* r = rec + it.first;
    uint2 rc[kSubPer];
    unsigned rank[kSubPer];
#pragma unroll
    for (int k = 0; k < kSubPer; ++k) {
        const unsigned j = k * kThreads + threadIdx.x;
        rc[k] = j < it.count ? r[j] : make_uint2(0xFFFFFFFFu, 0u);
    }
#pragma unroll
    for (int k = 0; k < kSubPer; ++k) {
        rank[k] = 0;
        if (rc[k].x != 0xFFFFFFFFu) rank[k] = atomicAdd(&hist[(rc[k].x >> kLcellBits) - tile0], 1u);
    }
    __syncthreads();
    const int per = (tps + kThreads - 1) / kThreads;            // <= 2
    const int lo = threadIdx.x * per, hi = min(lo + per, tps);
    unsigned s = 0;
    for (int i = lo; i < hi; ++i) s += hist[i];
    unsigned incl = s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned run = incl - s;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
    for (int i = lo; i < hi; ++i) {
        loff[i] = run;
        run += hist[i];
    }
    // reservations: tiles dealt to lanes interleaved (coalesced atomics)
    for (int i = threadIdx.x; i < tps; i += kThreads) {
        const unsigned c = hist[i];
        if (c) gbase[i] = atomicAdd(&cursor[tile0 + i], c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSubPer; ++k)
        if (rc[k].x != 0xFFFFFFFFu) stage[loff[(rc[k].x >> kLcellBits) - tile0] + rank[k]] = rc[k];
    __syncthreads();
    for (unsigned j = threadIdx.x; j < it.count; j += kThreads) {
        const uint2 q = stage[j];
        const unsigned t = (q.x >> kLcellBits) - tile0;
        out[gbase[t] + (j - loff[t])] = make_uint2(q.x & kLcellMask, q.y);
    }
}

// ---- pass C (Point glyph): fold a work item's records into an LDS tile, merge into the planes ---
//
// LDS atomics on gfx950 (tools/ubench_lds_atomics*.hip, measured): ds_add_f32 costs ~194 cycles
// per wave-instruction per CU whatever the addresses (lanes are serialized); ds_add_u32 / ds_max_i32
// ~7, ds_add_u64 ~12, ds_add_f64 ~22.  So the LDS copy of a tile is NOT float32:
//   sum    -> double, ds_add_f64  (also makes the tile sum exact to f32 precision)
//   weight -> u32,    ds_add_u32  (a Point-glyph weight is 1; exact)
//   max/min-> f32 bits, integer ds_max/ds_min (common.hpp)
// and is rounded to f32 once, when it is merged into the f32 state planes.
inline int tile_cell_bytes(unsigned mask) {
    return ((mask & 1) ? 8 : 0) + ((mask & 2) ? 4 : 0) + ((mask & 4) ? 4 : 0) + ((mask & 8) ? 4 : 0);
}

// LDS tile shape: 128 columns x as many rows (multiple of 8, <= 128) as fit ~150 KB of the CU's
// 160 KB LDS at the per-cell footprint of the requested planes.
inline BinGeom point_bin_geom(const GridDev& g, uint32_t mask, int row0, int rows) {
    BinGeom b;
    b.tile_w = 128;
    b.tile_h = std::min(128, (150 * 1024 / (std::max(tile_cell_bytes(mask), 4) * 128)) & ~7);
    b.bins_x = (g.W + b.tile_w - 1) / b.tile_w;
    b.bins_y = (rows + b.tile_h - 1) / b.tile_h;
    b.nbins = b.bins_x * b.bins_y;
    b.chunk = b.nbins <= 2048 ? 16384 : 8192;
    b.row0 = row0;
    b.rows = rows;
    b.sup_shift = 0;
    return b;
}

// FUSED (pcr_hip_engine_finalize_with_scatter): a launch that defines every cell of undefined planes also stores the finished
// bands -- finalize(rtype) of the cell where its reference tile is touched, NaN elsewhere -- from the tile it has in hand, so
// that the finalize pass does not read the planes back (C2: 134 MB).  The touched flags are complete: the counting pass of
// this scatter set them.  *done tells the finalize call whether the bands were stored (not when the scan split a bin).
template <unsigned MASK, bool FUSED>
__global__ void __launch_bounds__(kThreads)
k_tile_accum(GridDev g, BinGeom b, PlanesDev pl, const uint2* __restrict__ records,
             const BinItem* __restrict__ items, const unsigned* __restrict__ n_items, int fresh,
             FinalizeOuts fo, const uint32_t* __restrict__ touched, uint32_t* __restrict__ done) {
    extern __shared__ double lds_tile[];
    if (FUSED && blockIdx.x == 0 && threadIdx.x == 0) *done = (fresh == 2 && n_items[1] == 0u) ? 1u : 0u;
    if (blockIdx.x >= *n_items) return;
    // fresh: 0 the planes hold earlier contributions (read-modify-write); 1 they hold identity values (the merge stores where
    // the tile has something); 2 they are UNDEFINED and this launch has an item for every bin: every cell is stored, the
    // identity included -- unless the scan had to split a bin (n_items[1]), in which case k_fill_if has filled the planes
    // just before this launch and the merge proceeds as for 1 (a split bin's items merge with atomics).
    const bool full = fresh == 2 && n_items[1] == 0u;
    const BinItem it = items[blockIdx.x];
    const int cells = b.tile_w * b.tile_h;                     // multiple of 1024
    double* t_sum = lds_tile;
    unsigned* t_wgt = reinterpret_cast<unsigned*>(t_sum + ((MASK & 1) ? cells : 0));
    float* t_max = reinterpret_cast<float*>(t_wgt + ((MASK & 2) ? cells : 0));
    float* t_min = t_max + ((MASK & 4) ? cells : 0);

    // records: kUnroll independent 8-byte loads per lane, double-buffered -- the next batch is in flight
    // (32 KB per CU) while the current one goes through the dependent LDS atomics, and the first batch is
    // issued before the tile is even initialised.  Batches of 4 beat 8, 12 and 16 (0.142 vs 0.148 / 0.149 /
    // 0.152 ms on C2, same box, alternating runs): the kernel is not latency-bound, shorter batches interleave
    // the loads and the LDS atomics more finely.
    constexpr int kUnroll = 4;
    const uint2* rec = records + it.first;
    uint2 cur[kUnroll], nxt[kUnroll];
    auto fetch = [&](uint2 (&r)[kUnroll], unsigned j0) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const unsigned j = j0 + u * kThreads;
            r[u] = j < it.count ? stream_load(rec + j) : make_uint2(0xFFFFFFFFu, 0u);
        }
    };
    fetch(cur, threadIdx.x);

    for (int i = threadIdx.x; i < cells; i += kThreads) {      // identity fill
        if (MASK & 1) t_sum[i] = 0.0;
        if (MASK & 2) t_wgt[i] = 0u;
        if (MASK & 4) t_max[i] = -FLT_MAX;
        if (MASK & 8) t_min[i] = FLT_MAX;
    }
    __syncthreads();

    for (unsigned j0 = threadIdx.x; j0 < it.count; j0 += kUnroll * kThreads) {
        fetch(nxt, j0 + kUnroll * kThreads);                    // past the end: sentinels, no loads issued
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (cur[u].x == 0xFFFFFFFFu) continue;
            float val = __uint_as_float(cur[u].y);
            if (MASK & 1) unsafeAtomicAdd(&t_sum[cur[u].x], (double)val);
            if (MASK & 2) atomicAdd(&t_wgt[cur[u].x], 1u);
            if (MASK & 4) atomic_max_f32(&t_max[cur[u].x], val);
            if (MASK & 8) atomic_min_f32(&t_min[cur[u].x], val);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) cur[u] = nxt[u];
    }
    __syncthreads();

    // merge pass: tile -> HBM planes.  Exclusive owner => plain RMW; a split bin => atomics.
    const int bx = it.bin % b.bins_x, by = it.bin / b.bins_x;
    const int c0 = bx * b.tile_w, r0 = b.row0 + by * b.tile_h;        // r0 relative to the state window
    const int w = min(b.tile_w, g.W - c0), h = min(b.tile_h, b.row0 + b.rows - r0);
    const bool vec = !it.shared && (g.W % 4 == 0) && (w % 4 == 0) &&
                     ((((MASK & 1) ? reinterpret_cast<uintptr_t>(pl.sum) : 0) | ((MASK & 2) ? reinterpret_cast<uintptr_t>(pl.wgt) : 0) |
                       ((MASK & 4) ? reinterpret_cast<uintptr_t>(pl.mx) : 0) | ((MASK & 8) ? reinterpret_cast<uintptr_t>(pl.mn) : 0)) & 15) == 0;
    if (vec) {
        // one lane = 4 consecutive cells of a row: wide LDS reads, float4 global RMW; the planes'
        // loads of an iteration are independent and issued together
        const int qrow = b.tile_w >> 2;
        for (int i = threadIdx.x; i < qrow * h; i += kThreads) {
            int ly = i / qrow, lx = (i - ly * qrow) << 2;
            if (lx >= w) continue;
            int64_t cell = (int64_t)(r0 + ly) * g.W + (c0 + lx);
            int li = ly * b.tile_w + lx;
            float4 a1, a2, a4, a8, g1, g2, g4, g8;
            bool n1 = false, n2 = false, n4 = false, n8 = false;
            if (MASK & 1) {
                double2 lo = *reinterpret_cast<const double2*>(t_sum + li), hi = *reinterpret_cast<const double2*>(t_sum + li + 2);
                a1 = make_float4((float)lo.x, (float)lo.y, (float)hi.x, (float)hi.y);
                n1 = (lo.x != 0.0) | (lo.y != 0.0) | (hi.x != 0.0) | (hi.y != 0.0);
            }
            if (MASK & 2) {
                uint4 c = *reinterpret_cast<const uint4*>(t_wgt + li);
                a2 = make_float4((float)c.x, (float)c.y, (float)c.z, (float)c.w);
                n2 = (c.x | c.y | c.z | c.w) != 0u;
            }
            if (MASK & 4) { a4 = *reinterpret_cast<const float4*>(t_max + li); n4 = (a4.x != -FLT_MAX) | (a4.y != -FLT_MAX) | (a4.z != -FLT_MAX) | (a4.w != -FLT_MAX); }
            if (MASK & 8) { a8 = *reinterpret_cast<const float4*>(t_min + li); n8 = (a8.x != FLT_MAX) | (a8.y != FLT_MAX) | (a8.z != FLT_MAX) | (a8.w != FLT_MAX); }
            if (fresh) {
                // the planes hold their identity values (first scatter into them): nothing to read
                if (MASK & 1) g1 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MASK & 2) g2 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MASK & 4) g4 = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
                if (MASK & 8) g8 = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX);
            } else {
                if ((MASK & 1) && n1) g1 = *reinterpret_cast<const float4*>(pl.sum + cell);
                if ((MASK & 2) && n2) g2 = *reinterpret_cast<const float4*>(pl.wgt + cell);
                if ((MASK & 4) && n4) g4 = *reinterpret_cast<const float4*>(pl.mx + cell);
                if ((MASK & 8) && n8) g8 = *reinterpret_cast<const float4*>(pl.mn + cell);
            }
            // the Sum / Count planes leave with non-temporal stores: nothing reads them before the finalize pass, and
            // streaming 134 MB through the L2 only evicts the records still to be folded (C2 step -1.2 %)
            typedef float f4v __attribute__((ext_vector_type(4)));
            if (full) n1 = n2 = n4 = n8 = true;
            if ((MASK & 1) && n1) { g1.x += a1.x; g1.y += a1.y; g1.z += a1.z; g1.w += a1.w;
                __builtin_nontemporal_store(f4v{g1.x, g1.y, g1.z, g1.w}, reinterpret_cast<f4v*>(pl.sum + cell)); }
            if ((MASK & 2) && n2) { g2.x += a2.x; g2.y += a2.y; g2.z += a2.z; g2.w += a2.w;
                __builtin_nontemporal_store(f4v{g2.x, g2.y, g2.z, g2.w}, reinterpret_cast<f4v*>(pl.wgt + cell)); }
            if ((MASK & 4) && n4) { g4.x = fmaxf(g4.x, a4.x); g4.y = fmaxf(g4.y, a4.y); g4.z = fmaxf(g4.z, a4.z); g4.w = fmaxf(g4.w, a4.w); *reinterpret_cast<float4*>(pl.mx + cell) = g4; }
            if ((MASK & 8) && n8) { g8.x = fminf(g8.x, a8.x); g8.y = fminf(g8.y, a8.y); g8.z = fminf(g8.z, a8.z); g8.w = fminf(g8.w, a8.w); *reinterpret_cast<float4*>(pl.mn + cell) = g8; }
            if (FUSED && full) {
                // (the owned rows are the state window: a band cell has the plane cell's index)
                const float s4[4] = {g1.x, g1.y, g1.z, g1.w}, w4[4] = {g2.x, g2.y, g2.z, g2.w};
                const float x4[4] = {g4.x, g4.y, g4.z, g4.w}, m4[4] = {g8.x, g8.y, g8.z, g8.w};
                const int trow = ((g.st_r0 + r0 + ly) / g.th) * g.tiles_x;
                bool live[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) live[k] = touched[trow + (c0 + lx + k) / g.tw] != 0u;
                for (int o = 0; o < fo.n; ++o) {
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        v[k] = live[k] ? finalize_rt(fo.rtype[o], (MASK & 1) ? s4[k] : 0.f, (MASK & 2) ? w4[k] : 0.f,
                                                     (MASK & 4) ? x4[k] : -FLT_MAX, (MASK & 8) ? m4[k] : FLT_MAX) : NAN;
                    __builtin_nontemporal_store(f4v{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4v*>(fo.out[o] + cell));
                }
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < b.tile_w * h; i += kThreads) {
        int ly = i / b.tile_w, lx = i - ly * b.tile_w;
        if (lx >= w) continue;
        int64_t cell = (int64_t)(r0 + ly) * g.W + (c0 + lx);
        int li = ly * b.tile_w + lx;
        if (full) {                                    // (an item per bin, none of them shared)
            if (MASK & 1) pl.sum[cell] = (float)t_sum[li];
            if (MASK & 2) pl.wgt[cell] = (float)t_wgt[li];
            if (MASK & 4) pl.mx[cell] = t_max[li];
            if (MASK & 8) pl.mn[cell] = t_min[li];
        } else if (!it.shared) {
            if (MASK & 1) { double a = t_sum[li]; if (a != 0.0) pl.sum[cell] += (float)a; }
            if (MASK & 2) { unsigned a = t_wgt[li]; if (a) pl.wgt[cell] += (float)a; }
            if (MASK & 4) { float a = t_max[li]; if (a != -FLT_MAX) pl.mx[cell] = fmaxf(pl.mx[cell], a); }
            if (MASK & 8) { float a = t_min[li]; if (a != FLT_MAX) pl.mn[cell] = fminf(pl.mn[cell], a); }
        } else {
            if (MASK & 1) { double a = t_sum[li]; if (a != 0.0) atomic_add_f32(pl.sum + cell, (float)a); }
            if (MASK & 2) { unsigned a = t_wgt[li]; if (a) atomic_add_f32(pl.wgt + cell, (float)a); }
            if (MASK & 4) { float a = t_max[li]; if (a != -FLT_MAX) atomic_max_f32(pl.mx + cell, a); }
            if (MASK & 8) { float a = t_min[li]; if (a != FLT_MAX) atomic_min_f32(pl.mn + cell, a); }
        }
    }
}

// Undefined planes, and the scan found a bin it had to split (n_items[1]): such a bin's items merge with atomics, which need
// defined cells -- the planes get their identity values after all.  A no-op launch otherwise (the usual case).
__global__ void __launch_bounds__(256)
k_fill_if(const unsigned* __restrict__ n_items, PlanesDev pl, unsigned mask, int64_t cells4) {
    if (n_items[1] == 0u) return;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f), lo = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX),
                 hi = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cells4; i += stride) {
        if (mask & 1) reinterpret_cast<float4*>(pl.sum)[i] = zero;
        if (mask & 2) reinterpret_cast<float4*>(pl.wgt)[i] = zero;
        if (mask & 4) reinterpret_cast<float4*>(pl.mx)[i] = lo;
        if (mask & 8) reinterpret_cast<float4*>(pl.mn)[i] = hi;
    }
}

template <unsigned MASK>
void launch_accum(pcr_hip_engine* e, const GridDev& gd, const BinGeom& b, const PlanesDev& pl, const BinBuffers& bb,
                  bool fused = false) {
    size_t lds = (size_t)b.tile_w * b.tile_h * tile_cell_bytes(MASK);
    // fresh: only when every bin is owned by one workgroup of this launch can a store replace the read-modify-write
    auto go = [&](auto kernel, const FinalizeOuts& fo, uint32_t* done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3(bb.max_items), dim3(kThreads), lds, e->stream, gd, b, pl,
                           bb.records, bb.items, bb.n_items, e->planes_fresh, fo, (const uint32_t*)e->d_touched, done);
    };
    if (fused) go(&k_tile_accum<MASK, true>, e->fused_outs, e->fused_done);
    else go(&k_tile_accum<MASK, false>, FinalizeOuts{}, nullptr);
}

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace

namespace pcrhip {

int bin_points(pcr_hip_engine* e, const GridDev& gd, const BinGeom& b, const double* x, const double* y, const float* v,
               uint64_t n, RecordKind kind, const GlyphDev* gl, unsigned item_records, BinBuffers* out, bool every_bin) {
    const int max_items = b.nbins + (int)(n / item_records) + 1;
    const size_t rec_bytes = sizeof(uint2);

    // Counts and cursors per VIRTUAL XCD (blockIdx % 8; workgroups are dealt to the XCDs round-robin): a bin's record range is
    // split into eight sub-ranges and a (workgroup, bin) run of ~21 records only ever shares its first and last 128-byte line
    // with runs written through the same L2, where they merge -- shared between XCDs they left as partial lines (504 MB
    // written for 400 MB of records, profiles/r02_C2_rocprof.md).  Both passes use the same block -> points mapping.
    constexpr int nvx = kVirtualXcds;
    size_t off = 0;                                             // scratch carve-up
    const size_t o_count = off;  off += align256((size_t)nvx * b.nbins * 4);
    const size_t o_cursor = off; off += align256((size_t)nvx * b.nbins * 4);
    const size_t o_nitems = off; off += 256;
    const size_t o_items = off;  off += align256((size_t)max_items * sizeof(BinItem));
    const size_t o_rec = off;    off += align256((size_t)n * rec_bytes);
    const size_t o_keys = off;   off += align256((size_t)n * sizeof(unsigned));
    int rc = ensure_scratch(e, off);
    if (rc) return rc;
    char* s = e->d_scratch;
    unsigned* d_count = reinterpret_cast<unsigned*>(s + o_count);
    unsigned* d_cursor = reinterpret_cast<unsigned*>(s + o_cursor);
    unsigned* d_nitems = reinterpret_cast<unsigned*>(s + o_nitems);
    BinItem* d_items = reinterpret_cast<BinItem*>(s + o_items);
    unsigned* d_keys = reinterpret_cast<unsigned*>(s + o_keys);

    PCR_HIP_TRY(hipMemsetAsync(d_count, 0, (size_t)nvx * b.nbins * 4, e->stream));
    {
        ScopedKernelTimer t(e, "k_bin_count");
        // the count pass walks the cloud in the scatter pass's blocks (its chunks, then 4096-point blocks for the ragged end)
        BinGeom bc = b;
        const ScatterShape sh = scatter_shape(b.nbins);
        bc.chunk = sh.chunk();
        const int full_blocks = scatter_full_blocks(sh, v, n, kind == RecordKind::Index);
        const uint64_t done = (uint64_t)full_blocks * sh.chunk();
        const int cb = count_blocks(b.nbins, full_blocks, e->num_cus);
        const unsigned tail_blocks = (unsigned)((n - done + 4095) / 4096);
        if (cb > 1) {
            const unsigned cblocks = (unsigned)nvx * (((unsigned)(full_blocks + nvx - 1) / nvx + cb - 1) / cb) + tail_blocks;
            launch_bin_count<true>(gd.tiles_x * gd.tiles_y == 1, dim3(cblocks), (size_t)b.nbins * 4, e->stream,
                                   gd, bc, (unsigned)full_blocks, cb, nvx, x, y, n, d_keys, d_count, e->d_touched, e->d_counters);
        } else {
            const int split = count_split(bc.chunk, b.nbins);
            launch_bin_count<false>(gd.tiles_x * gd.tiles_y == 1, dim3((unsigned)full_blocks * (unsigned)split + tail_blocks),
                                    (size_t)b.nbins * 4, e->stream,
                                    gd, bc, (unsigned)full_blocks, split, nvx, x, y, n, d_keys, d_count, e->d_touched, e->d_counters);
        }
    }
    {
        ScopedKernelTimer t(e, "k_bin_scan");
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(kThreads), 0, e->stream, b.nbins, nvx, item_records, d_count,
                           d_cursor, d_items, d_nitems, every_bin ? 1 : 0);
    }
    {
        uint2* d_rec = reinterpret_cast<uint2*>(s + o_rec);
        if (kind == RecordKind::Index) launch_bin_scatter<true>(e, b, nvx, d_keys, v, n, d_cursor, d_rec);
        else launch_bin_scatter<false>(e, b, nvx, d_keys, v, n, d_cursor, d_rec);
        out->records = d_rec;
    }
    PCR_HIP_TRY(hipGetLastError());
    out->items = d_items;
    out->n_items = d_nitems;
    out->max_items = max_items;
    return PCR_HIP_OK;
}

// Two-level counting sort for the Point glyph on grids with more LDS tiles than one pass can count
// (16384^2 = 21 888 tiles): level 1 groups the points by runs of 2^s consecutive tiles (about sqrt(tiles)
// groups), level 2 sorts every group by tile.  Both levels stream; see k_sub_count / k_sub_scatter.
int two_level_shift(const pcr_hip_engine* e, int tiles) {
    // tile kMaxTiles-1 with local cell 2^15-1 would encode to the dropped-point sentinel 0xFFFFFFFF
    if (!e->two_level || tiles >= kMaxTiles) return 0;
    int s = 1;
    while ((1 << (2 * s)) < tiles) ++s;                                  // groups ~ tiles per group ~ sqrt(tiles)
    while (((tiles + (1 << s) - 1) >> s) > e->max_bins) ++s;
    return (1 << s) <= kMaxSubBins ? s : 0;
}

int bin_points_two_level(pcr_hip_engine* e, const BinGeom& tiles, const double* x, const double* y, const float* v,
                         uint64_t n, bool index_records, unsigned item_records, BinBuffers* out, bool every_bin) {
    BinGeom l1 = tiles;                                                   // first level: groups of tiles
    l1.nbins = (tiles.nbins + (1 << tiles.sup_shift) - 1) >> tiles.sup_shift;
    const ScatterShape sh1 = scatter_shape(l1.nbins);
    l1.chunk = sh1.chunk();                                               // the count pass walks the scatter pass's blocks (bin_points)
    constexpr int nvx = kVirtualXcds;                                     // (measured neutral here: the first level's runs are ~170 records long)
    const unsigned sub_records = kSubPer * kThreads;
    const int full_blocks = scatter_full_blocks(sh1, v, n, index_records);
    const unsigned blocks = (unsigned)full_blocks + (unsigned)((n - (uint64_t)full_blocks * sh1.chunk() + 4095) / 4096);
    const int max_items1 = l1.nbins + (int)(n / sub_records) + 1;
    const int max_items2 = tiles.nbins + (int)(n / item_records) + 1;

    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
    const size_t o_count1 = carve((size_t)nvx * l1.nbins * 4), o_cursor1 = carve((size_t)nvx * l1.nbins * 4), o_nitems1 = carve(8);
    const size_t o_items1 = carve((size_t)max_items1 * sizeof(BinItem));
    const size_t o_count2 = carve((size_t)tiles.nbins * 4), o_cursor2 = carve((size_t)tiles.nbins * 4), o_nitems2 = carve(8);
    const size_t o_items2 = carve((size_t)max_items2 * sizeof(BinItem));
    const size_t o_keys = carve((size_t)n * 4), o_rec1 = carve((size_t)n * 8), o_rec2 = carve((size_t)n * 8);
    int rc = ensure_scratch(e, off);
    if (rc) return rc;
    char* s = e->d_scratch;
    auto U = [&](size_t o) { return reinterpret_cast<unsigned*>(s + o); };
    BinItem* d_items1 = reinterpret_cast<BinItem*>(s + o_items1);
    BinItem* d_items2 = reinterpret_cast<BinItem*>(s + o_items2);
    uint2* d_rec1 = reinterpret_cast<uint2*>(s + o_rec1);
    uint2* d_rec2 = reinterpret_cast<uint2*>(s + o_rec2);

    PCR_HIP_TRY(hipMemsetAsync(U(o_count1), 0, (size_t)nvx * l1.nbins * 4, e->stream));
    PCR_HIP_TRY(hipMemsetAsync(U(o_count2), 0, (size_t)tiles.nbins * 4, e->stream));
    {
        ScopedKernelTimer t(e, "k_bin_count");
        const int cb = count_blocks(l1.nbins, full_blocks, e->num_cus);
        const unsigned tail_blocks = blocks - (unsigned)full_blocks;
        if (cb > 1) {
            const unsigned cblocks = (unsigned)nvx * (((unsigned)(full_blocks + nvx - 1) / nvx + cb - 1) / cb) + tail_blocks;
            launch_bin_count<true>(e->gd.tiles_x * e->gd.tiles_y == 1, dim3(cblocks), (size_t)l1.nbins * 4, e->stream,
                                   e->gd, l1, (unsigned)full_blocks, cb, nvx, x, y, n, U(o_keys), U(o_count1), e->d_touched, e->d_counters);
        } else {
            const int split = count_split(l1.chunk, l1.nbins);
            launch_bin_count<false>(e->gd.tiles_x * e->gd.tiles_y == 1, dim3((unsigned)full_blocks * (unsigned)split + tail_blocks),
                                    (size_t)l1.nbins * 4, e->stream,
                                    e->gd, l1, (unsigned)full_blocks, split, nvx, x, y, n, U(o_keys), U(o_count1), e->d_touched, e->d_counters);
        }
    }
    {
        ScopedKernelTimer t(e, "k_bin_scan");
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(kThreads), 0, e->stream, l1.nbins, nvx, sub_records, U(o_count1),
                           U(o_cursor1), d_items1, U(o_nitems1), 0);
    }
    if (index_records) launch_bin_scatter<true>(e, l1, nvx, U(o_keys), v, n, U(o_cursor1), d_rec1);
    else launch_bin_scatter<false>(e, l1, nvx, U(o_keys), v, n, U(o_cursor1), d_rec1);
    const int tps = 1 << tiles.sup_shift;
    {
        ScopedKernelTimer t(e, "k_sub_count");
        hipLaunchKernelGGL(k_sub_count, dim3(max_items1), dim3(kThreads), (size_t)tps * 4, e->stream, tiles.sup_shift,
                           d_rec1, d_items1, U(o_nitems1), U(o_count2));
    }
    {
        ScopedKernelTimer t(e, "k_bin_scan");
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(kThreads), 0, e->stream, tiles.nbins, 1, item_records, U(o_count2),
                           U(o_cursor2), d_items2, U(o_nitems2), every_bin ? 1 : 0);
    }
    {
        ScopedKernelTimer t(e, "k_sub_scatter");
        const size_t lds = (size_t)sub_records * sizeof(uint2) + (size_t)tps * 4 * 3;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sub_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_sub_scatter, dim3(max_items1), dim3(kThreads), lds, e->stream, tiles.sup_shift, d_rec1, d_items1,
                           U(o_nitems1), U(o_cursor2), d_rec2);
    }
    PCR_HIP_TRY(hipGetLastError());
    out->records = d_rec2;
    out->items = d_items2;
    out->n_items = U(o_nitems2);
    out->max_items = max_items2;
    return PCR_HIP_OK;
}

// Bands of state rows, each with at most kMaxBins LDS tiles; every band is a full pass over the points
// (routing keys are cheap: 5 ps per point and band) that only keeps the points of its rows.
static int point_bands(const pcr_hip_engine* e, uint32_t mask, int* band_rows) {
    const GridDev& g = e->gd;
    const BinGeom b = point_bin_geom(g, mask, 0, g.st_rows);
    *band_rows = band_rows_for(g, b.tile_w, b.tile_h, e->max_bins);
    if (*band_rows <= 0) return 0;
    return (g.st_rows + *band_rows - 1) / *band_rows;
}

bool binned_point_supported(const pcr_hip_engine* e, uint32_t mask) {
    if (mask == 0 || (mask & ~15u)) return false;
    int band_rows = 0;
    const int nbands = point_bands(e, mask, &band_rows);
    const bool two_level = nbands != 1 && two_level_shift(e, point_bin_geom(e->gd, mask, 0, e->gd.st_rows).nbins) > 0;
    if (!two_level && (nbands < 1 || nbands > kMaxBands)) return false;
    // not worth the fixed cost of sweeping every tile for a handful of points
    uint64_t cells = (uint64_t)e->gd.W * e->gd.st_rows;
    if (e->forced_path != 2 && e->stats.points_in * 16 < cells) return false;
    return e->stats.points_in < (1ull << 32) - (1ull << 20);
}

int binned_point(pcr_hip_engine* e, uint32_t mask, const PlanesDev& pl,
                 const double* x, const double* y, const float* v, uint64_t n) {
    int band_rows = 0;
    int nbands = point_bands(e, mask, &band_rows);
    int total_bins = 0;
    BinGeom b = point_bin_geom(e->gd, mask, 0, e->gd.st_rows);
    const int shift = nbands != 1 ? two_level_shift(e, b.nbins) : 0;
    // Undefined planes (pcr_hip_engine_planes_fresh(e, 2)): one band of bins covers the whole state window, so the tile
    // pass can define every cell itself -- an item for every bin, every cell stored -- and the state initialisation costs
    // no pass of its own.  Needs whole float4 groups per plane row (the merge's vector form is per row, the scalar form
    // covers the rest); anything else (two sort levels, several bands) fills the planes first.
    const int64_t cells = (int64_t)e->gd.st_rows * e->gd.W;
    const bool define_all = e->planes_fresh == 2 && (nbands == 1 || shift > 0) && cells % 4 == 0 &&
                            ((reinterpret_cast<uintptr_t>(pl.sum) | reinterpret_cast<uintptr_t>(pl.wgt) |
                              reinterpret_cast<uintptr_t>(pl.mx) | reinterpret_cast<uintptr_t>(pl.mn)) & 15) == 0;
    if (e->planes_fresh == 2 && !define_all) {
        int rc = fill_identity(e, mask, pl);
        if (rc) return rc;
        e->planes_fresh = 1;
    }
    if (shift > 0) {                                            // one sweep, two sort levels
        b.sup_shift = shift;
        BinBuffers bb{};
        int rc = bin_points_two_level(e, b, x, y, v, n, false, kPointItemRecords, &bb, define_all);
        if (rc) return rc;
        if (define_all)
            hipLaunchKernelGGL(k_fill_if, dim3(2048), dim3(256), 0, e->stream, bb.n_items, pl, mask, cells / 4);
        const bool fused = define_all && e->fused_outs.n > 0 && e->fused_done && e->gd.W % 4 == 0 &&
                           e->gd.own_r0 == e->gd.st_r0 && e->gd.own_r1 - e->gd.own_r0 == e->gd.st_rows;
        e->fused_taken = fused;
        ScopedKernelTimer t(e, "k_tile_accum");
        switch (mask) {
#define PCR_ACC(M) case M: launch_accum<M>(e, e->gd, b, pl, bb, fused); break;
            PCR_ACC(1) PCR_ACC(2) PCR_ACC(3) PCR_ACC(4) PCR_ACC(5) PCR_ACC(6) PCR_ACC(7) PCR_ACC(8)
            PCR_ACC(9) PCR_ACC(10) PCR_ACC(11) PCR_ACC(12) PCR_ACC(13) PCR_ACC(14) PCR_ACC(15)
#undef PCR_ACC
            default: return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_point: empty plane mask");
        }
        total_bins = b.nbins;
        nbands = 0;
    } else if (nbands < 1) {
        return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_point: grid cannot be binned");
    }
    for (int band = 0; band < nbands; ++band) {
        const int row0 = band * band_rows, rows = std::min(band_rows, e->gd.st_rows - row0);
        GridDev gd = e->gd;                                     // this band's points only
        gd.own_r0 = std::max(e->gd.own_r0, e->gd.st_r0 + row0);
        gd.own_r1 = std::min(e->gd.own_r1, e->gd.st_r0 + row0 + rows);
        if (gd.own_r0 >= gd.own_r1) continue;
        b = point_bin_geom(e->gd, mask, row0, rows);
        total_bins += b.nbins;
        BinBuffers bb{};
        int rc = bin_points(e, gd, b, x, y, v, n, RecordKind::Value, nullptr, kPointItemRecords, &bb, define_all);
        if (rc) return rc;
        if (define_all)
            hipLaunchKernelGGL(k_fill_if, dim3(2048), dim3(256), 0, e->stream, bb.n_items, pl, mask, cells / 4);
        // the bands too, when the caller asked (pcr_hip_engine_finalize_with_scatter): this launch stores every cell of the
        // window from float4 groups (define_all), and a band cell has the plane cell's index when the owned rows are the window
        const bool fused = define_all && e->fused_outs.n > 0 && e->fused_done && e->gd.W % 4 == 0 &&
                           e->gd.own_r0 == e->gd.st_r0 && e->gd.own_r1 - e->gd.own_r0 == e->gd.st_rows;
        e->fused_taken = fused;
        ScopedKernelTimer t(e, "k_tile_accum");
        switch (mask) {
#define PCR_ACC(M) case M: launch_accum<M>(e, gd, b, pl, bb, fused); break;
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
    e->stats.num_bins = total_bins;
    return PCR_HIP_OK;
}

}  // namespace pcrhip
