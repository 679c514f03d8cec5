// scatter_moments.hip -- large-sigma Gaussian glyphs without per-cell scatter.
//
// A default-sigma, unrotated Gaussian footprint is separable, and its dependence on the point's
// sub-cell offset s (s' = s - 1/2 in [-1/2, 1/2)) can be expanded exactly enough:
//
//   exp(-(d - s)^2 / 2 sigma^2) = exp(-u^2/2 sigma^2) * exp(u s'/sigma^2) * exp(-s'^2/2 sigma^2),   u = d - 1/2
//                               = sum_k A_k(d) m_k(s'),   A_k(d) = exp(-u^2/2 sigma^2) c_k(u/sigma^2),
//                                                         m_k(s') = T_k(2 s') exp(-s'^2/2 sigma^2)
//
// where c_k(z) are the Chebyshev coefficients of s' -> exp(z s') on [-1/2, 1/2] (c_k ~ 2 (z/4)^k / k!: they
// fall 2^(k-1) times faster than the Taylor terms (z/2)^k / k!, and |T_k| <= 1 makes the tail bound rigorous).
// |z| <= (r + 1/2)/sigma^2 is small for sigma >= 2 cells, so a total order K of 3 (sigma=16) to 9 (sigma=2)
// reproduces every weight to <= 5e-5 relative (rigorous bound in make_plan; half the 1e-4 test tolerance).  Then
//
//   splat = sum_{k+l<=K} (A_k (x) B_l) * M_kl ,   M_kl[cell] = sum_{points centred in cell} v m_k(s'x) n_l(s'y)
//
// i.e. (1) ONE Point-style pass builds P = (K+1)(K+2)/2 moment planes per plane kind -- points are
// binned by tile, sorted by cell inside LDS, and reduced per cell in registers, no atomics on the
// moments at all -- and (2) separable (2r+1)-tap convolutions, independent of the number of points,
// spread them: column pass U_k = sum_l B_l * M_kl, row pass out += sum_k A_k * U_k.  The footprint
// window |d| <= r is the convolution support; clipping to the reference tile of the centre cell (Q4)
// becomes "taps never cross a reference-tile edge".  The 1e-6 weight cut-off of the reference is
// provably dead for these parameters (checked on the host), otherwise the path is not taken.
//
// Work per point drops from (2r+1)^2 weighted atomics (9409 at sigma=16) to ~P multiply-adds.
// Points the expansion cannot represent (non-finite value, centre cell not the routed cell at a
// grid edge) are painted by the wave-per-point direct kernel afterwards.
#include "bin16.hpp"

#include <type_traits>
#include <vector>

using namespace pcrhip;

namespace {

constexpr int kMomThreads = 256;      // per-tile moment reduction: four waves, three workgroups per CU
constexpr int kPad = 20;              // zero padding of the tap tables (16-output windows, 4 source rows per step)
constexpr int kTileW = 64, kTileH = 16;                  // a wave folds one row of the tile at a time; ~3 K records staged in LDS
constexpr int kSortChunk = 3840;                         // records staged per round (45 KB of LDS: three workgroups per CU)
constexpr int kMaxK = 9;
constexpr double kTruncationBound = 5e-5;      // make_plan: rigorous bound on the relative error of any weight

struct MomPlan {
    BinGeom bins;
    int K, P, r;
    float inv2sx2, inv2sy2;           // 1 / (2 sigma^2), cells^-2
    float inv_csx_f, inv_csy_f;
};

__device__ __forceinline__ bool finite_f(float v) { return (__float_as_uint(v) & 0x7F800000u) != 0x7F800000u; }

// ---- binning: the shared front-end (bin16.hpp) on records {local cell, value, s'x, s'y} -----------------------
// s' = the reference's f32 sub-cell offset, recentred to [-1/2, 1/2).  A point whose value is not finite, or whose
// centre cell is not the routed cell (grid edge), cannot be represented by moments: it goes to the list.
typedef float pcr_f2 __attribute__((ext_vector_type(2)));

struct MomentMaker {
    static constexpr bool kVectorGeometry = false;     // k_b16_scatter: grid and bin geometry in vector registers (bin16.hpp)
    static constexpr bool kCentre = true;
    static constexpr bool kOwnsX = false;
    static constexpr bool kFixup = false;
    static constexpr int kPer = 16, kBatch = 16;
    struct Chan {};
    __device__ __forceinline__ Chan load(uint64_t) const { return Chan{}; }
    __device__ __forceinline__ bool make(const GridDev&, const BinGeom&, const b16::Routed16&, const PointGeom& pg, float val,
                                         const Chan&, uint4& rec) const {
        if (!finite_f(val)) return false;
        rec.y = __float_as_uint(val);
        rec.z = __float_as_uint((float)(pg.fcx - floor(pg.fcx)) - 0.5f);
        rec.w = __float_as_uint((float)(pg.fcy - floor(pg.fcy)) - 0.5f);
        return true;
    }
};

// ---- per-tile moments: the tile's records are sorted by cell INSIDE LDS (the records themselves are staged, 12
// bytes each), then a lane owns a cell and folds its records into P (x2) moments held in registers -- no atomics
// on the moments, and no second trip to memory: round 2 kept only 16-bit positions in LDS and gathered the 16-byte
// records by position, 128-byte lines for 16 useful bytes (5.5 GB fetched for 0.8 GB of records,
// profiles/r02_moments_split.md).  Tiles are 64 x 16 cells so that a tile's ~3 K records fit three workgroups per CU;
// a crowded tile is folded in rounds of kSortChunk records.  One workgroup per tile, empty tiles included, so that
// every moment cell is written and the planes need no memset.
template <int K, unsigned MASK>
__global__ void __launch_bounds__(kMomThreads)
k_tile_moments(GridDev g, BinGeom b, float inv2sx2, float inv2sy2, const uint4* __restrict__ records,
               const unsigned* __restrict__ bin_start, float* __restrict__ mom_v, float* __restrict__ mom_w,
               int64_t plane_stride) {
    constexpr int P = (K + 1) * (K + 2) / 2;
    constexpr int kCells = kTileW * kTileH;                                 // 1024
    constexpr int kPer = kCells / kMomThreads;                              // 4 cells per thread
    constexpr int kRecs = (kSortChunk + kMomThreads - 1) / kMomThreads;     // 15 records per thread and round
    __shared__ unsigned off[kCells + 1];
    __shared__ float sv[kSortChunk], ssx[kSortChunk], ssy[kSortChunk];
    __shared__ unsigned wave_tot[kMomThreads / 64];

    const int bin = blockIdx.x;
    const unsigned first = bin_start[bin], count = bin_start[bin + 1] - first;
    const int bx = bin % b.bins_x, by = bin / b.bins_x;
    const int c0 = bx * kTileW, r0 = by * kTileH;
    const int w = min(kTileW, g.W - c0), h = min(kTileH, g.st_rows - r0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    for (unsigned cbase = 0; cbase == 0 || cbase < count; cbase += kSortChunk) {
        const unsigned cn = min((unsigned)kSortChunk, count - cbase);
        const uint4* rec = records + first + cbase;
        uint4 rc[kRecs];
#pragma unroll
        for (int k = 0; k < kRecs; ++k) {
            const unsigned j = threadIdx.x + k * kMomThreads;
            rc[k] = j < cn ? stream_load(rec + j) : make_uint4(b16::kNullCell, 0u, 0u, 0u);
        }
        for (int i = threadIdx.x; i <= kCells; i += kMomThreads) off[i] = 0;
        __syncthreads();
        // counting sort by cell: the counting atomic returns the record's rank inside its cell
        unsigned rk[kRecs];
#pragma unroll
        for (int k = 0; k < kRecs; ++k) {
            rk[k] = 0;
            if (rc[k].x < (unsigned)kCells) rk[k] = atomicAdd(&off[rc[k].x], 1u);
        }
        __syncthreads();
        {
            const int lo = threadIdx.x * kPer;
            unsigned c[kPer], s = 0;
#pragma unroll
            for (int q = 0; q < kPer; ++q) { c[q] = off[lo + q]; s += c[q]; }
            unsigned incl = s;
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
            for (int q = 0; q < kPer; ++q) { off[lo + q] = run; run += c[q]; }
            if (threadIdx.x == kMomThreads - 1) off[kCells] = run;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kRecs; ++k) {
            if (rc[k].x >= (unsigned)kCells) continue;
            const unsigned pos = off[rc[k].x] + rk[k];
            sv[pos] = __uint_as_float(rc[k].y);
            ssx[pos] = __uint_as_float(rc[k].z);
            ssy[pos] = __uint_as_float(rc[k].w);
        }
        __syncthreads();

        // a wave folds one row of the tile at a time: lane = cell, 256-byte plane stores.  (Ordering the cells by record
        // count so that a wave's 64 cells run the same number of rounds was tried: the fold got ~2x shorter and the plane
        // stores, no longer row segments, cost 3x that: k_tile_moments 0.54 -> 1.20 ms at K = 3, profiles/r03_moments.md.)
        for (int ly = wave; ly < kTileH; ly += kMomThreads / 64) {
            const int lx = lane;
            if (ly >= h) break;                                             // wave-uniform
            const int cell = ly * kTileW + lx;
            const unsigned e0 = off[cell];
            const unsigned cnt = min(off[cell + 1] - e0, (unsigned)kSortChunk);   // (the bound only guards the loop)
            pcr_f2 acc[P];                                                  // {sum of v m, sum of m}
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p] = pcr_f2{0.f, 0.f};
            // every lane runs every round; a lane past its cell's last record folds a null record (weights 0): straight-line
            // code, no select per accumulator
            for (unsigned e = 0; __any(e < cnt); ++e) {
                const bool live = e < cnt;
                const unsigned idx = live ? e0 + e : 0u;
                const float val = live ? sv[idx] : 0.f, sx = ssx[idx], sy = ssy[idx];
                float mx[K + 1], ny[K + 1];
                // m_k = T_k(2 s') * envelope: T_0 = 1, T_1 = x, T_(k+1) = 2 x T_k - T_(k-1), x = 2 s' in [-1, 1)
                const float x2 = 2.0f * sx, y2 = 2.0f * sy;
                mx[0] = live ? expf(-(sx * sx) * inv2sx2) : 0.f;
                ny[0] = expf(-(sy * sy) * inv2sy2);
                mx[1] = mx[0] * x2;
                ny[1] = ny[0] * y2;
#pragma unroll
                for (int k = 2; k <= K; ++k) {
                    mx[k] = 2.0f * x2 * mx[k - 1] - mx[k - 2];
                    ny[k] = 2.0f * y2 * ny[k - 1] - ny[k - 2];
                }
                // {av, aw}[p] += {val, 1} * m_kl as ONE packed fused multiply-add: the fold is bound by vector-instruction
                // issue (a wave64 instruction holds its SIMD for four cycles, packed or not: profiles/r03_gauss1_sq.md)
                const pcr_f2 v1{val, 1.0f};
                int p = 0;
#pragma unroll
                for (int k = 0; k <= K; ++k) {
#pragma unroll
                    for (int l = 0; l <= K - k; ++l) {
                        const float m = mx[k] * ny[l];
                        acc[p] = __builtin_elementwise_fma(v1, pcr_f2{m, m}, acc[p]);
                        ++p;
                    }
                }
            }
            if (lx >= w) continue;
            const int64_t gcell = (int64_t)(r0 + ly) * g.W + (c0 + lx);
            if (cbase == 0) {
                // written once, read by the column pass much later: streamed past the L2
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (MASK & 1) __builtin_nontemporal_store(acc[p].x, mom_v + p * plane_stride + gcell);
                    if (MASK & 2) __builtin_nontemporal_store(acc[p].y, mom_w + p * plane_stride + gcell);
                }
            } else if (cnt > 0) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (MASK & 1) mom_v[p * plane_stride + gcell] += acc[p].x;
                    if (MASK & 2) mom_w[p * plane_stride + gcell] += acc[p].y;
                }
            }
        }
        __syncthreads();
    }
}

// ---- convolutions ---------------------------------------------------------------------------------------
// Both passes keep 16 consecutive outputs per lane in registers and slide the (2r+1)-tap window over
// 16 + 2r sources; a source feeds up to 16 fused multiply-adds, taps are wave-uniform scalar operands
// (zero-padded tables, s_load).  Sources come from LDS so that neighbouring output blocks share them
// instead of re-reading L2 (the column pass was L2-bound at 7x read amplification without it).
// Output group g (4 outputs) only meets non-zero taps for source steps mm in [4g, 4g + 2r + 3]: the
// other (zero-tap) steps are skipped with a scalar branch.
// One step = 4 consecutive sources v0..v3 against the 20 taps T(0..19) = tw[-3..16] of the step's window:
// output j gains T(j + 3 - u) * v_u.  The FMAs are issued as packed pairs (v_pk_fma_f32) whose tap operands
// are scalar register pairs; a pair must start at an even register, which holds for outputs (j, j+1), j even,
// only against the odd sources -- so the even sources accumulate into a second file B shifted by one output
// (B[i] = output i - 1; its two end entries are never read), and the two files are added at the end.
// Every step fetches its 20 taps with two scalar loads (the tables sit in the scalar cache); other waves of
// the SIMD cover that latency.  GM = output groups (4 outputs each) that can meet a non-zero tap.
typedef pcr_f2 pcr_f2_u __attribute__((aligned(4)));          // a tap pair at any float offset

struct ConvAcc {
    pcr_f2 a[8];       // a[p] = outputs (2p, 2p + 1), fed by the odd sources
    pcr_f2 b[9];       // b[p] = outputs (2p - 1, 2p), fed by the even sources; b[0].x and b[8].y are never read
};

__device__ __forceinline__ void conv_clear(ConvAcc& c) {
#pragma unroll
    for (int p = 0; p < 8; ++p) c.a[p] = pcr_f2{0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 9; ++p) c.b[p] = pcr_f2{0.f, 0.f};
}

__device__ __forceinline__ float conv_out(const ConvAcc& c, int j) {
    return (j & 1) ? c.a[j >> 1].y + c.b[(j >> 1) + 1].x : c.a[j >> 1].x + c.b[j >> 1].y;
}

// t[i] = taps (T(2i), T(2i+1)), T(0..19) = tw[-3..16]
template <unsigned GM>
__device__ __forceinline__ void conv_step(ConvAcc& c, const pcr_f2 (&t)[10], float v0, float v1, float v2, float v3) {
    const pcr_f2 s0{v0, v0}, s1{v1, v1}, s2{v2, v2}, s3{v3, v3};
#pragma unroll
    for (int p = 0; p < 8; ++p) {                      // outputs 2p, 2p + 1
        if (!(GM & (1u << (p / 2)))) continue;
        c.a[p] = __builtin_elementwise_fma(t[p + 1], s1, c.a[p]);
        c.a[p] = __builtin_elementwise_fma(t[p], s3, c.a[p]);
    }
#pragma unroll
    for (int p = 0; p < 9; ++p) {                      // outputs 2p - 1, 2p
        const bool on = (p >= 1 && (GM & (1u << ((2 * p - 1) / 4)))) || (p <= 7 && (GM & (1u << ((2 * p) / 4))));
        if (!on) continue;
        c.b[p] = __builtin_elementwise_fma(t[p + 1], s0, c.b[p]);
        c.b[p] = __builtin_elementwise_fma(t[p], s2, c.b[p]);
    }
}

// The whole sweep of one lane over its 16 + 2r sources.  src(c) yields source c (c = 0 is r sources before
// output 0); tap of output j for source c: tbase[j - c].  Output group gq meets non-zero taps only for steps
// mm in [4 gq, 4 gq + 2r + 3]: three ramp-up and three ramp-down steps run with fewer groups.
template <typename Src>
__device__ __forceinline__ void conv_sweep(ConvAcc& c, int r, const float* __restrict__ tbase, Src src) {
    auto step = [&](auto gm, int mm) {
        pcr_f2 t[10];
        const pcr_f2_u* __restrict__ tw = reinterpret_cast<const pcr_f2_u*>(tbase - mm - 3);
#pragma unroll
        for (int i = 0; i < 10; ++i) t[i] = tw[i];
        conv_step<decltype(gm)::value>(c, t, src(mm), src(mm + 1), src(mm + 2), src(mm + 3));
    };
    using std::integral_constant;
    // r >= 6 (make_plan): the three ramps and the steady range do not overlap
    const int steady_hi = (2 * r + 3) & ~3;             // all four groups inside for mm in [12, steady_hi]
    step(integral_constant<unsigned, 1>{}, 0);
    step(integral_constant<unsigned, 3>{}, 4);
    step(integral_constant<unsigned, 7>{}, 8);
    // walked downwards: the tap pointer then climbs by 4 per step and every scalar load is pointer + immediate
    for (int mm = steady_hi; mm >= 12; mm -= 4) step(integral_constant<unsigned, 15>{}, mm);
    step(integral_constant<unsigned, 14>{}, steady_hi + 4);
    step(integral_constant<unsigned, 12>{}, steady_hi + 8);
    if (steady_hi + 12 < 16 + 2 * r) step(integral_constant<unsigned, 8>{}, steady_hi + 12);
}

// Column pass: U_k[y][x] = sum_{l <= K-k} sum_dy B_l(dy) M_kl[y - dy][x]; sources stay inside the output
// row's reference tile and the state window.  A workgroup owns 64 columns x 64 output rows (one lane per
// column, 16 rows per wave) and stages the 64 + 2r source rows of one moment plane at a time in LDS.
// NPF > 0: every thread keeps its NPF rows of the NEXT plane in flight (registers) behind the sweep of
// the current one, so that HBM/L2 latency is paid once per workgroup, not once per plane; NPF == 0 is
// the plain version for windows too tall for that.
template <int NPF>
__global__ void __launch_bounds__(256)
k_conv_col(GridDev g, int K, int r, int yblocks_per_tile, const float* __restrict__ taps_y,
           const float* __restrict__ mom, int64_t plane_stride, float* __restrict__ u_out) {
    extern __shared__ float lds_f[];                   // [64 + 2r + 4][64]
    const int lane = threadIdx.x & 63;
    // wave-uniform by construction; readfirstlane makes it an SGPR so that the taps below are
    // fetched with scalar loads and feed the FMAs as scalar operands
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int x = blockIdx.x * 64 + lane;
    const int k = blockIdx.z;
    const int trow = blockIdx.y / yblocks_per_tile, yb = blockIdx.y - trow * yblocks_per_tile;
    // reference tile rows, in window coordinates
    const int t_lo = max(trow * g.th - g.st_r0, 0), t_hi = min(min((trow + 1) * g.th, g.H) - g.st_r0, g.st_rows);
    const int Y0 = t_lo + yb * 64;
    if (Y0 >= t_hi) return;                            // whole workgroup
    const int nsrc = 64 + 2 * r, nalloc = nsrc + 4;    // LDS row s <-> window row Y0 - r + s; 4 zero rows of slack
    const int tap_w = 2 * r + 1 + 2 * kPad;
    ConvAcc acc;
    conv_clear(acc);
    // pair index of (k, 0): sum_{i<k} (K + 1 - i)
    int p0 = 0;
    for (int i = 0; i < k; ++i) p0 += K + 1 - i;
    const bool xin = x < g.W;
    const int xc = xin ? x : 0;
    const bool active = Y0 + wave * 16 < t_hi;
    const float* tb = lds_f + wave * 16 * 64 + lane;

    // Staging: row pointers are wave-uniform (scalar), the lane only adds its column.  Lanes right of the
    // grid read column 0 and carry throw-away results, so no load is predicated per lane.
    float pre[NPF > 0 ? NPF : 1];
    // (the wave index goes through an opaque asm so that the per-row scalar state -- pointers, validity -- is
    //  recomputed at each use instead of being hoisted out of the plane loop as ~100 live scalars)
    auto issue = [&](int l) {                          // wave w owns rows w, w+4, ...
        int w = wave;
        asm volatile("" : "+s"(w));
        const int u_lo = max(0, (t_lo - (Y0 - r) - w + 3) >> 2);                 // rows of the reference tile ...
        const int u_hi = (min(t_hi - (Y0 - r), nsrc) - w + 3) >> 2;              // ... and of the window: u in [u_lo, u_hi)
        const float* __restrict__ rowp = mom + (int64_t)(p0 + l) * plane_stride + (int64_t)(Y0 - r + w) * g.W;
        const int64_t step4 = (int64_t)4 * g.W;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            pre[u] = (u >= u_lo && u < u_hi) ? rowp[xc] : 0.f;
            rowp += step4;
        }
    };
    if (NPF > 0) issue(0);
    for (int l = 0; l <= K - k; ++l) {
        __syncthreads();                               // the previous plane's readers are done
        if (NPF > 0) {
            int w = wave;
            asm volatile("" : "+s"(w));
            const int u_end = (nalloc - w + 3) >> 2;
            float* dst = lds_f + w * 64 + lane;
#pragma unroll
            for (int u = 0; u < NPF; ++u)
                if (u < u_end) dst[u * 256] = pre[u];
        } else {
            const float* __restrict__ plane = mom + (int64_t)(p0 + l) * plane_stride;
            for (int s0 = wave; s0 < nalloc; s0 += 32) {
                float vals[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int sr = s0 + 4 * u, yy = Y0 - r + sr;
                    const float* __restrict__ rowp = plane + (int64_t)yy * g.W;
                    vals[u] = (sr < nsrc && yy >= t_lo && yy < t_hi) ? rowp[xc] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (s0 + 4 * u < nalloc) lds_f[(s0 + 4 * u) * 64 + lane] = vals[u];
            }
        }
        __syncthreads();
        if (NPF > 0 && l < K - k) issue(l + 1);
        const float* __restrict__ tl = taps_y + l * tap_w + kPad + 2 * r;      // wave-uniform
        conv_sweep(acc, r, tl, [&](int c) { return tb[c * 64]; });       // source c = LDS row 16 * wave + c
    }
    if (!xin || !active) return;
    float* uo = u_out + (int64_t)k * plane_stride;
    const int y0 = Y0 + wave * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (y0 + j < t_hi) uo[(int64_t)(y0 + j) * g.W + x] = conv_out(acc, j);
}

// The column pass on the matrix cores.  The sweep of one wave -- 16 output rows x 64 columns against its
// 16 + 2r source rows -- is a banded Toeplitz product: out[i][j] = sum_c T[i][c] * src[c][j], T[i][c] = tap(i - c).
// v_mfma_f32_16x16x4_f32 takes f32 operands and accumulates in f32 (no precision to give up) at the packed-FMA
// rate, but needs NO vector-ALU instruction per product: per step of four source rows a lane reads one tap and
// four sources from LDS and issues four MFMAs (the four 16-column blocks share the tap operand).  The zero band
// of T costs (16 + 2r + 3) / (2r + 1) - 1 = 23 % extra products at r = 48; the vector sweep above reached 43 % of
// the FMA rate.  Operand lane maps (one f32 per lane): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// C[row = 4 (lane >> 4) + reg][col = lane & 15].
// LDS rows are 80 floats apart: the four source rows a B operand touches then fall on disjoint banks.
typedef float pcr_f4 __attribute__((ext_vector_type(4)));
constexpr int kColStride = 80;

// Staging (NI > 0; needs W % 4 == 0): a lane moves FOUR consecutive columns of one row per load (16-byte global
// load, 16-byte LDS store), a wave four rows per instruction -- NI = ceil(rows / 16) loads per lane and plane, kept
// in flight in registers behind the sweep of the previous plane.  The first version staged one float per lane with
// wave-uniform row pointers and validity (as k_conv_col does): ~670 scalar instructions per plane and wave, as much
// issue time as the MFMAs themselves (profiles/r02_conv_mfma.md).  NI == 0: plain float loop, any width.
// NW waves per workgroup = 16 NW output rows x 64 columns.  Four waves stage 64 + 2r rows for 64 outputs; eight stage
// 128 + 2r for 128: at r = 48 that is 7 staging loads per wave and plane instead of 11, and a SIMD hosts four sweeping
// waves (two workgroups of eight) instead of three -- the waits this kernel has are on the CU's one texture-address path,
// where every 1-KB load instruction takes ~190 cycles to issue in the burst after a barrier
// (profiles/r03_conv_col_phases.md).
template <int NI, int NW>
__global__ void __launch_bounds__(64 * NW)
k_conv_col_mfma(GridDev g, int K, int r, int yblocks_per_tile, const float* __restrict__ taps_y,
                const float* __restrict__ mom, int64_t plane_stride, float* __restrict__ u_out) {
    constexpr int kT = 64 * NW, kRowsWg = 16 * NW, kRowsIt = 4 * NW;      // threads; output rows; rows one staging round covers
    extern __shared__ float lds_f[];                   // [16 (NW - 1) + 4 steps][kColStride] sources | (K + 1) tap tables
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int x = blockIdx.x * 64 + lane;
    const int trow = blockIdx.y / yblocks_per_tile, yb = blockIdx.y - trow * yblocks_per_tile;
    const int t_lo = max(trow * g.th - g.st_r0, 0), t_hi = min(min((trow + 1) * g.th, g.H) - g.st_r0, g.st_rows);
    const int Y0 = t_lo + yb * kRowsWg;
    if (Y0 >= t_hi) return;                            // whole workgroup
    const int steps = (16 + 2 * r + 3) >> 2;           // four source rows per step; rows past 16 + 2r meet zero taps
    const int nsrc = kRowsWg + 2 * r, nalloc = kRowsWg - 16 + 4 * steps;   // LDS row s <-> window row Y0 - r + s; the sweep of the
                                                            // last wave ends at row 16 (NW - 1) + 4 steps - 1 (0..3 zero rows of slack)
    const int tap_w = 2 * r + 1 + 2 * kPad;
    float* lds_taps = lds_f + nalloc * kColStride;
    for (int i = threadIdx.x; i < (K + 1) * tap_w; i += kT) lds_taps[i] = taps_y[i];
    pcr_f4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = pcr_f4{0.f, 0.f, 0.f, 0.f};
    const bool xin = x < g.W;
    const bool active = Y0 + wave * 16 < t_hi;
    const int j16 = lane & 15, kg = lane >> 4;

    // staging map: LDS row 4 NW it + 4 wave + kg, columns 4 j16 .. 4 j16 + 3
    const int srow0 = 4 * wave + kg, c4 = 4 * j16;
    const bool cin = blockIdx.x * 64 + c4 < g.W;       // W % 4 == 0: the four columns are inside together
    const int lo_s = t_lo - (Y0 - r), hi_s = min(t_hi - (Y0 - r), nsrc);       // valid LDS rows [lo_s, hi_s)
    pcr_f4 pre[NI > 0 ? NI : 1];
    auto issue = [&](int pair) {                       // moment planes are stored in the order the pairs are visited
        const float* __restrict__ base = mom + (int64_t)pair * plane_stride + (int64_t)(Y0 - r + srow0) * g.W +
                                         (blockIdx.x * 64 + c4);
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int sr = srow0 + kRowsIt * it;
            const bool ok = cin && sr >= lo_s && sr < hi_s;
            pre[it] = ok ? *reinterpret_cast<const pcr_f4*>(base + (int64_t)kRowsIt * it * g.W) : pcr_f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // One workgroup walks ALL pairs (k, l), l = 0..K-k, of its 64 x 64 outputs: the load of the next plane is always
    // in flight behind the current sweep (one exposed memory latency per workgroup, not one per k).
    const int npairs = (K + 1) * (K + 2) / 2;
    if (NI > 0) issue(0);
    int k = 0, l = 0;
    for (int pair = 0; pair < npairs; ++pair) {
        __syncthreads();                               // the previous plane's readers (and the U_k hand-over) are done
        if (NI > 0) {
            float* dst = lds_f + srow0 * kColStride + c4;
#pragma unroll
            for (int it = 0; it < NI; ++it)
                if (srow0 + kRowsIt * it < nalloc) *reinterpret_cast<pcr_f4*>(dst + kRowsIt * it * kColStride) = pre[it];
        } else {
            const float* __restrict__ plane = mom + (int64_t)pair * plane_stride;
            const int xc = xin ? x : 0;
            for (int s0 = wave; s0 < nalloc; s0 += 8 * NW) {
                float vals[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int sr = s0 + NW * u, yy = Y0 - r + sr;
                    const float* __restrict__ rowp = plane + (int64_t)yy * g.W;
                    vals[u] = (sr < nsrc && yy >= t_lo && yy < t_hi) ? rowp[xc] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (s0 + NW * u < nalloc) lds_f[(s0 + NW * u) * kColStride + lane] = vals[u];
            }
        }
        __syncthreads();
        if (NI > 0 && pair + 1 < npairs) issue(pair + 1);
        if (active) {
            // output row i = j16 of this wave, source c = 4 s + kg (LDS row 16 wave + c): tap index i - c
            const float* ta = lds_taps + l * tap_w + kPad + 2 * r + j16 - kg;
            const float* sb = lds_f + (16 * wave + kg) * kColStride + j16;
            // Software-pipelined over two operand sets: the reads of step s + 1 are issued before the MFMAs of step s.
            // (The index goes through an opaque asm and the loaded set is "used" by an empty asm after the MFMAs --
            // otherwise the compiler folds the prefetch back into "load, wait, use" at the top of the next iteration.)
            struct Ops { float a, b0, b1, b2, b3; };
            auto load = [&](int sidx) {
                int sn = min(sidx, steps - 1);                      // past the end: re-read the last step (unused)
                asm volatile("" : "+s"(sn));
                const float* row = sb + 4 * sn * kColStride;
                return Ops{ta[-4 * sn], row[0], row[16], row[32], row[48]};
            };
            auto mac = [&](const Ops& o) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a, o.b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a, o.b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a, o.b2, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a, o.b3, acc[3], 0, 0, 0);
            };
            auto pin = [](Ops& o) {
                __builtin_amdgcn_sched_barrier(0);                  // after the MFMAs above, not in their middle
                asm volatile("" : "+v"(o.a), "+v"(o.b0), "+v"(o.b1), "+v"(o.b2), "+v"(o.b3));
            };
            Ops p = load(0), q;
            for (int s4 = 0; s4 < steps; s4 += 2) {
                q = load(s4 + 1);
                __builtin_amdgcn_sched_barrier(0);                  // reads first, then the MFMAs that cover their latency
                mac(p);
                pin(q);
                p = load(s4 + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (s4 + 1 < steps) mac(q);
                pin(p);
            }
        }
        if (l < K - k) { ++l; continue; }
        // U_k is complete.  It leaves through LDS (rows 16 wave .. 16 wave + 15 of the staging area, free once every
        // wave has finished its sweep): an accumulator holds 4 rows x 16 columns per lane group, the planes want
        // whole 256-byte row segments per store.
        __syncthreads();
        if (active) {
            float* ob = lds_f + (16 * wave + 4 * kg) * kColStride + j16;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ob[q * kColStride + cb * 16] = acc[cb][q];
                acc[cb] = pcr_f4{0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (xin) {
                float* uo = u_out + (int64_t)k * plane_stride;
                const int y0 = Y0 + wave * 16;
                const float* rb = lds_f + 16 * wave * kColStride + lane;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (y0 + j < t_hi) uo[(int64_t)(y0 + j) * g.W + x] = rb[j * kColStride];
            }
        }
        ++k;
        l = 0;
    }
}

// Row pass + accumulate: out[y][x] += sum_k sum_dx A_k(dx) U_k[y][x - dx], sources inside the output
// column's reference tile.  One lane owns 16 consecutive outputs of a row; a wave owns RW rows x 16*LPR
// columns (LPR lanes per row) and keeps its source rows in a wave-private LDS strip -- no workgroup
// barriers.  Logical column c of a strip lives at c + c/16: lanes 16 columns apart then hit different
// banks (row strides are chosen = 64/RW mod 64 so that the RW rows of a wave do not collide either).
// NPR > 0: the NPR x 64 columns of each row of the NEXT plane are kept in flight in registers behind
// the sweep of the current one; NPR == 0: plain version for wider windows.
template <int LPR_SHIFT, int NPR>
__global__ void __launch_bounds__(256)
k_conv_row_accum(GridDev g, int K, int r, int xunits_per_tile, int rs, const float* __restrict__ taps_x,
                 const float* __restrict__ u_in, int64_t plane_stride, float* __restrict__ out_plane, int store) {
    constexpr int LPR = 1 << LPR_SHIFT, RW = 64 / LPR, COLS = 16 * LPR;
    extern __shared__ float lds_f[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tcol = blockIdx.x / xunits_per_tile, xu = blockIdx.x - tcol * xunits_per_tile;
    const int t_lo = tcol * g.tw, t_hi = min((tcol + 1) * g.tw, g.W);
    const int X0 = t_lo + xu * COLS;
    const int Y0 = (blockIdx.y * 4 + wave) * RW;
    if (X0 >= t_hi || Y0 >= g.st_rows) return;        // wave-uniform; this kernel has no workgroup barrier
    const int span = COLS + 2 * r;                     // source columns [X0 - r, X0 + COLS + r)
    const int spanp = span + 4;                        // + zero slack read by the last step
    float* wt = lds_f + wave * (RW * rs);
    const int rr = lane >> LPR_SHIFT, xb = lane & (LPR - 1);
    const float* lb = wt + rr * rs + 17 * xb;
    const int tap_w = 2 * r + 1 + 2 * kPad;
    ConvAcc acc;
    conv_clear(acc);

    constexpr int NP = NPR > 0 ? RW * NPR : 1;
    float pre[NP];
    // valid source columns of the strip, as strip-relative c in [c_lo, c_hi)
    const int c_lo = max(t_lo - (X0 - r), 0), c_hi = min(t_hi - (X0 - r), span);
    // (lane goes through an opaque asm in the staging code so that per-load masks and offsets are recomputed
    //  where used instead of being hoisted out of the plane loop as dozens of live scalar pairs)
    auto issue = [&](int k) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const float* uk = u_in + (int64_t)k * plane_stride + (X0 - r);
#pragma unroll
        for (int row = 0; row < RW; ++row) {
            const int gy = Y0 + row;
            const float* __restrict__ urow = uk + (int64_t)min(gy, g.st_rows - 1) * g.W;      // scalar
            const unsigned width = gy < g.st_rows ? (unsigned)(c_hi - c_lo) : 0u;
#pragma unroll
            for (int u = 0; u < NPR; ++u) {
                const int c = ln + 64 * u;
                pre[row * (NPR > 0 ? NPR : 1) + u] = (unsigned)(c - c_lo) < width ? urow[c] : 0.f;
            }
        }
    };
    if (NPR > 0) issue(0);
    for (int k = 0; k <= K; ++k) {
        if (NPR > 0) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            float* wb = wt + ln + (ln >> 4);                  // column c = ln + 64u lives at c + c/16 = wb + 68u
#pragma unroll
            for (int row = 0; row < RW; ++row)
#pragma unroll
                for (int u = 0; u < NPR; ++u)
                    if (ln + 64 * u < spanp) wb[row * rs + 68 * u] = pre[row * (NPR > 0 ? NPR : 1) + u];
        } else {
            const float* uk = u_in + (int64_t)k * plane_stride;
            for (int row = 0; row < RW; ++row) {
                const int gy = Y0 + row;
                const float* urow = uk + (int64_t)min(gy, g.st_rows - 1) * g.W;
                for (int c0 = lane; c0 < spanp; c0 += 512) {         // eight coalesced loads in flight
                    float vals[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = c0 + 64 * u, gx = X0 - r + c;
                        vals[u] = (c < span && gx >= t_lo && gx < t_hi && gy < g.st_rows) ? urow[gx] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = c0 + 64 * u;
                        if (c < spanp) wt[row * rs + c + (c >> 4)] = vals[u];
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (NPR > 0 && k < K) issue(k + 1);
        const float* __restrict__ tk = taps_x + k * tap_w + kPad + 2 * r;
        conv_sweep(acc, r, tk, [&](int c) { return lb[c + (c >> 4)]; });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // results return through the strip for a coalesced read-modify-write
    float* ob = wt + rr * rs + 17 * xb;
#pragma unroll
    for (int j = 0; j < 16; ++j) ob[j] = conv_out(acc, j);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int row = 0; row < RW; ++row) {
        const int gy = Y0 + row;
        if (gy >= g.st_rows) break;
        for (int c = lane; c < COLS; c += 64) {
            const int gx = X0 + c;
            if (gx < t_hi) {
                const float a = wt[row * rs + c + (c >> 4)];
                // store: the plane is still UNDEFINED and this launch visits every cell of it once (one window = the whole
                // state window): the row pass is the plane's initialisation, zeros included
                if (store) out_plane[(int64_t)gy * g.W + gx] = a;
                else if (a != 0.f) out_plane[(int64_t)gy * g.W + gx] += a;
            }
        }
    }
}

// ---- fallback: wave-per-point direct splat of the listed points --------------------------------------
template <unsigned MASK>
struct DirectSink {
    const GridDev& g;
    PlanesDev pl;
    __device__ __forceinline__ void add(int row, int col, float vw, float w) {
        int64_t cell = (int64_t)(row - g.st_r0) * g.W + col;
        if (MASK & 1) atomic_add_f32(pl.sum + cell, vw);
        if (MASK & 2) atomic_add_f32(pl.wgt + cell, w);
    }
};

template <unsigned MASK>
__global__ void __launch_bounds__(256)
k_gauss_list(GridDev g, GlyphDev gl, PlanesDev pl, const unsigned* __restrict__ list,
             const unsigned* __restrict__ count, const double* __restrict__ x, const double* __restrict__ y,
             const float* __restrict__ v) {
    const unsigned n = *count;
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    DirectSink<MASK> sink{g, pl};
    for (unsigned base = wave * 64; base < n; base += nwaves * 64) {
        unsigned j = base + lane;
        bool valid = j < n;
        GaussParams q{};
        if (valid) {
            uint64_t i = list[j];
            PointGeom pg = point_geom(g, x[i], y[i]);
            valid = pg.valid;
            if (valid) q = gauss_params(g, gl, pg, v[i], load_chan(gl, i));
        }
        unsigned long long todo = __ballot(valid);
        while (todo) {
            int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            GaussParams u = lane_bcast(q, src);
            gauss_splat_wave(u, lane, sink);
        }
    }
}

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// tiles of the window `g` describes
void plan_bins(MomPlan& p, const GridDev& g) {
    p.bins.tile_w = kTileW;
    p.bins.tile_h = kTileH;
    p.bins.bins_x = (g.W + kTileW - 1) / kTileW;
    p.bins.bins_y = (g.st_rows + kTileH - 1) / kTileH;
    p.bins.nbins = p.bins.bins_x * p.bins.bins_y;
    p.bins.chunk = 0;                                    // set where the engine is known (moments_gauss)
    p.bins.row0 = 0;
    p.bins.rows = g.st_rows;
    p.bins.sup_shift = 0;
}

// Chebyshev coefficients c_0..c_kmax of s' -> exp(z s') on [-1/2, 1/2] (= x -> exp(z x / 2) on [-1, 1]) by
// Gauss-Chebyshev quadrature on 64 nodes (exact to rounding for this entire function).
void cheb_coeffs(double z, int kmax, double* c) {
    constexpr int N = 64, kTab = 41;
    static double node[N], cs[kTab][N];
    static const bool ready = [] {
        const double pi = 3.14159265358979323846;
        for (int j = 0; j < N; ++j) {
            node[j] = std::cos(pi * (j + 0.5) / N);
            for (int k = 0; k < kTab; ++k) cs[k][j] = std::cos(pi * k * (j + 0.5) / N);
        }
        return true;
    }();
    (void)ready;
    double f[N];
    for (int j = 0; j < N; ++j) f[j] = std::exp(0.5 * z * node[j]);
    for (int k = 0; k <= kmax && k < kTab; ++k) {
        double a = 0.0;
        for (int j = 0; j < N; ++j) a += f[j] * cs[k][j];
        c[k] = (k == 0 ? 1.0 : 2.0) * a / N;
    }
}

// host: plan from the glyph spec; false when the expansion does not apply
bool make_plan(const GridDev& g, const GlyphDev& gl, MomPlan* out) {
    if (gl.type != PCR_HIP_GLYPH_GAUSSIAN || gl.sigma_x || gl.sigma_y || gl.rotation) return false;
    if (gl.def_rotation != 0.0f) return false;                       // cos == 1, sin == 0 exactly
    const float sx = gl.def_sigma_x * (float)g.inv_csx, sy = gl.def_sigma_y * (float)g.inv_csy;
    const float R = std::min(3.0f * std::max(sx, sy), gl.max_radius);
    if (!(sx > 0.0f) || sy == 0.0f || !(R > 5.0f) || R > 200.0f) return false;   // r >= 6: smaller footprints are cheap to splat
    const int r = (int)std::ceil(R);
    const double asy = std::fabs((double)sy), sx2 = (double)sx * sx, sy2 = asy * asy;
    // the reference drops weights < 1e-6 (glyph_kernels.cu:166): must never trigger inside the window
    const double qmax = 0.5 * ((r + 1.0) * (r + 1.0) / sx2 + (r + 1.0) * (r + 1.0) / sy2);
    if (qmax > 13.5) return false;
    // Total order K: the dropped terms are sum_{k+l>K} c_k(zx) T_k c_l(zy) T_l, |T| <= 1, so relative to the true
    // factor exp(zx s'x + zy s'y) >= exp(-(zx+zy)/2) the error of a weight is <= e^((zx+zy)/2) sum_{k+l>K} |c_k||c_l|,
    // largest at the rim of the footprint (zx = (r+1/2)/sx^2, where the weight itself is ~1 % of the peak).  K is the
    // smallest instantiated order whose bound is <= kTruncationBound = 5e-5.  Every weight of a cell's sum is positive and
    // carries at most that relative error, so the sum does too: half of the 1e-4 tolerance every Gaussian path is
    // tested to, the other half being ~20x the float32 accumulation noise.  (Round 1 asked for 1e-5: order 4 at
    // sigma = 16, whose bound is 3e-7 -- fifteen moment planes where ten do: order 3 bounds at 1.6e-5.)
    const double zx = (r + 0.5) / sx2, zy = (r + 0.5) / sy2;
    constexpr int kTail = 40;
    double cx[kTail + 1], cy[kTail + 1];
    cheb_coeffs(zx, kTail, cx);
    cheb_coeffs(zy, kTail, cy);
    int K = -1;
    for (int cand : {3, 4, 5, 6, 7, 9}) {
        double tail = 0.0;
        for (int k = 0; k <= kTail; ++k)
            for (int l = 0; l <= kTail; ++l)
                if (k + l > cand) tail += std::fabs(cx[k]) * std::fabs(cy[l]);
        if (std::exp(0.5 * (zx + zy)) * tail <= kTruncationBound) { K = cand; break; }
    }
    if (K < 0) return false;
    MomPlan p;
    p.K = K;
    p.P = (K + 1) * (K + 2) / 2;
    p.r = r;
    p.inv2sx2 = (float)(1.0 / (2.0 * sx2));
    p.inv2sy2 = (float)(1.0 / (2.0 * sy2));
    plan_bins(p, g);
    *out = p;
    return true;
}

// A grid whose window has more moment tiles than one binning pass takes is processed in row bands: each band is a
// complete run of the path on a WINDOW of the state (the band's rows plus r rows on either side, where its points'
// footprints can land), with the points whose centre row lies in the band.  Moment and U planes only ever cover
// one window (16384^2 at order 6: 17 GB per band instead of 68 GB), and the windows' outputs add up in the state.
struct MomBand {
    int own_r0, own_r1;       // rows (grid coordinates) whose points this band takes
    int win_r0, win_rows;     // the band's window, in rows of the engine's state window
};

bool plan_bands(const pcr_hip_engine* e, const MomPlan& p, std::vector<MomBand>* bands) {
    const GridDev& g = e->gd;
    const int bins_x = (g.W + kTileW - 1) / kTileW;
    const int max_tile_rows = b16::max_bins(e) / bins_x;
    const int rows_total = g.st_rows;
    if ((rows_total + kTileH - 1) / kTileH <= max_tile_rows) {                  // one band: the whole window
        if (bands) bands->push_back({g.own_r0, g.own_r1, 0, rows_total});
        return true;
    }
    const int band_rows = max_tile_rows * kTileH - 2 * p.r;
    if (band_rows < kTileH) return false;
    const int nbands = (rows_total + band_rows - 1) / band_rows;
    if (nbands > kMaxBands) return false;
    for (int b0 = 0; b0 < rows_total && bands; b0 += band_rows) {
        const int b1 = std::min(rows_total, b0 + band_rows);
        MomBand mb;
        mb.own_r0 = std::max(g.own_r0, g.st_r0 + b0);
        mb.own_r1 = std::min(g.own_r1, g.st_r0 + b1);
        mb.win_r0 = std::max(0, b0 - p.r);
        mb.win_rows = std::min(rows_total, b1 + p.r) - mb.win_r0;
        if (mb.own_r0 < mb.own_r1) bands->push_back(mb);
    }
    return true;
}

void fill_taps(std::vector<float>& t, int K, int r, double s2) {
    const int tap_w = 2 * r + 1 + 2 * kPad;
    t.assign((size_t)(K + 1) * tap_w, 0.0f);
    double c[kMaxK + 1];
    for (int d = -r; d <= r; ++d) {
        const double u = d - 0.5;
        cheb_coeffs(u / s2, K, c);
        const double g = std::exp(-u * u / (2.0 * s2));
        for (int k = 0; k <= K; ++k) t[(size_t)k * tap_w + kPad + d + r] = (float)(g * c[k]);
    }
}

template <int K, unsigned MASK>
void launch_moments(pcr_hip_engine* e, const GridDev& gw, const MomPlan& p, const uint4* rec, const unsigned* bin_start,
                    float* mom_v, float* mom_w, int64_t stride) {
    hipLaunchKernelGGL((k_tile_moments<K, MASK>), dim3(p.bins.nbins), dim3(kMomThreads), 0, e->stream, gw, p.bins,
                       p.inv2sx2, p.inv2sy2, rec, bin_start, mom_v, mom_w, stride);
}

template <unsigned MASK>
void dispatch_moments(pcr_hip_engine* e, const GridDev& gw, const MomPlan& p, const uint4* rec, const unsigned* bin_start,
                      float* mom_v, float* mom_w, int64_t stride) {
    if (p.K == 3) launch_moments<3, MASK>(e, gw, p, rec, bin_start, mom_v, mom_w, stride);
    else if (p.K == 4) launch_moments<4, MASK>(e, gw, p, rec, bin_start, mom_v, mom_w, stride);
    else if (p.K == 5) launch_moments<5, MASK>(e, gw, p, rec, bin_start, mom_v, mom_w, stride);
    else if (p.K == 6) launch_moments<6, MASK>(e, gw, p, rec, bin_start, mom_v, mom_w, stride);
    else if (p.K == 7) launch_moments<7, MASK>(e, gw, p, rec, bin_start, mom_v, mom_w, stride);
    else launch_moments<9, MASK>(e, gw, p, rec, bin_start, mom_v, mom_w, stride);
}

}  // namespace

namespace pcrhip {

bool moments_supported(const pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask) {
    if (mask == 0 || (mask & ~3u)) return false;
    MomPlan p;
    if (!make_plan(e->gd, gl, &p) || !plan_bands(e, p, nullptr)) return false;
    if (e->stats.points_in >= (1ull << 32) - (1ull << 20)) return false;
    if (e->forced_path == 3) return true;
    // worth it when painting footprints costs more than the point-count independent convolutions
    // (measured on MI355X, 4096^2, 50 M points: LDS-tile splat ~1.5 ps per cell update; convolutions ~0.09 ps
    //  per cell x pair x tap for both plane kinds; moment passes 37 ps + 0.75 ps x pairs per point)
    const double n = (double)e->stats.points_in;
    const double splat = n * (2.0 * p.r + 1) * (2.0 * p.r + 1) * 1.5e-12;
    const double conv = (double)e->gd.W * e->gd.st_rows * p.P * (2.0 * p.r + 1) * 0.9e-13 +
                        n * (37e-12 + 0.75e-12 * p.P);
    return conv < splat;
}

int moments_gauss(pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask, const PlanesDev& pl,
                  const double* x, const double* y, const float* v, uint64_t n, bool planes_undefined) {
    MomPlan p;
    std::vector<MomBand> bands;
    if (!make_plan(e->gd, gl, &p) || !plan_bands(e, p, &bands))
        return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_glyph: moment path not applicable");
    // Undefined planes (pcr_hip_engine_planes_fresh(e, 2)): when ONE window covers the whole state window, the row pass
    // visits every cell of a plane exactly once and can store instead of accumulating -- the state initialisation costs no
    // pass and no plane read; several windows add up in the state, which then has to hold its identity values first.
    const bool store = planes_undefined && bands.size() == 1 && bands[0].win_r0 == 0 && bands[0].win_rows == e->gd.st_rows;
    if (planes_undefined && !store) {
        int frc = fill_identity(e, mask, pl);
        if (frc) return frc;
    }
    const int kinds = ((mask & 1) ? 1 : 0) + ((mask & 2) ? 1 : 0);
    const int tap_w = 2 * p.r + 1 + 2 * kPad;
    int max_rows = 0, max_bins = 0;
    for (const MomBand& mb : bands) {
        max_rows = std::max(max_rows, mb.win_rows);
        max_bins = std::max(max_bins, p.bins.bins_x * ((mb.win_rows + kTileH - 1) / kTileH));
    }
    const int64_t max_cells = (int64_t)e->gd.W * max_rows;

    // scratch: the front-end's buffers, then the moment and U planes of one window
    const b16::Layout L = b16::layout(0, max_bins, n, (unsigned)kSortChunk);
    size_t off = L.end;
    const size_t o_mom = off;    off += align256((size_t)kinds * p.P * max_cells * 4);
    const size_t o_u = off;      off += align256((size_t)(p.K + 1) * max_cells * 4);
    int rc = ensure_scratch(e, off);
    if (rc) return rc;
    char* s = e->d_scratch;
    unsigned* d_fbc = reinterpret_cast<unsigned*>(s + L.o_fbc);
    unsigned* d_fbl = reinterpret_cast<unsigned*>(s + L.o_fbl);
    float* d_mom = reinterpret_cast<float*>(s + o_mom);
    float* d_u = reinterpret_cast<float*>(s + o_u);

    // tap tables: x taps then y taps, resident on the device while the glyph spec stays the same
    const float sx = gl.def_sigma_x * (float)e->gd.inv_csx, sy = gl.def_sigma_y * (float)e->gd.inv_csy;
    const float* d_taps = nullptr;
    size_t ntaps = 0;
    rc = shared_taps(e, p.K, p.r, sx, sy,
                     [](std::vector<float>& t, int K, int r, float tsx, float tsy) {
                         std::vector<float> ty;
                         fill_taps(t, K, r, (double)tsx * tsx);
                         fill_taps(ty, K, r, (double)tsy * tsy);
                         t.insert(t.end(), ty.begin(), ty.end());
                     },
                     &d_taps, &ntaps);
    if (rc) return rc;
    const float* taps_x = d_taps;
    const float* taps_y = d_taps + ntaps / 2;

    // shapes that do not depend on the band
    const GridDev& ge = e->gd;
    const int yblocks = (std::min(ge.th, ge.H) + 63) / 64;
    const size_t col_lds = (size_t)(64 + 2 * p.r + 4) * 64 * sizeof(float);
    const int col_rows_per_wave = (64 + 2 * p.r + 4 + 3) / 4;
    const int col_rows_mfma = 48 + 4 * ((16 + 2 * p.r + 3) >> 2);       // three workgroups per CU at r = 48, K = 3
    const size_t col_lds_mfma = ((size_t)col_rows_mfma * kColStride + (size_t)(p.K + 1) * tap_w) * sizeof(float);
    // eight waves (128 output rows) per workgroup where two such workgroups fit a CU and the reference tile has the rows
    const int col_rows_mfma8 = col_rows_mfma + 64;
    const size_t col_lds_mfma8 = ((size_t)col_rows_mfma8 * kColStride + (size_t)(p.K + 1) * tap_w) * sizeof(float);
    int col_nw = (col_lds_mfma8 <= (size_t)80 * 1024 && std::min(ge.th, ge.H) >= 128) ? 8 : 4;
    if (const char* t = std::getenv("PCR_HIP_CONV_WAVES")) {            // experiments: force the workgroup shape
        const int v = std::atoi(t);
        if (v == 4 || (v == 8 && col_lds_mfma8 <= (size_t)160 * 1024)) col_nw = v;
    }
    const int yblocks8 = (std::min(ge.th, ge.H) + 127) / 128;
    // row pass shape: the strip width (1024 / 256 / 64 columns) that wastes the fewest lanes on this tile width
    const int eff_tw = std::min(ge.tw, ge.W);
    int lpr_shift = 6;
    double best = -1.0;
    for (int sh : {6, 4, 2}) {
        const int cols = 16 << sh;
        const double util = (double)eff_tw / ((double)((eff_tw + cols - 1) / cols) * cols);
        if (util > best + 0.05) { best = util; lpr_shift = sh; }
    }
    const int rw = 64 >> lpr_shift, cols = 16 << lpr_shift;
    const int spanp = cols + 2 * p.r + 4;
    const int need = spanp + (spanp >> 4) + 1, bank_off = (64 / rw) % 64;
    const int rs = ((need - bank_off + 63) / 64) * 64 + bank_off;      // rs >= need, rs = 64/RW (mod 64)
    const size_t row_lds = (size_t)4 * rw * rs * sizeof(float);
    const int xunits = (eff_tw + cols - 1) / cols;
    const int npr_need = (spanp + 63) / 64;                             // 64-column loads per row of a strip

    int total_bins = 0;
    for (const MomBand& mb : bands) {
        // the band's window as a grid of its own: every kernel below indexes rows relative to it
        GridDev g = ge;
        g.own_r0 = mb.own_r0;
        g.own_r1 = mb.own_r1;
        g.st_r0 = ge.st_r0 + mb.win_r0;
        g.st_rows = mb.win_rows;
        MomPlan pb = p;
        plan_bins(pb, g);
        pb.bins.chunk = b16::chunk_of<MomentMaker>();
        if (pb.bins.nbins > max_bins) return fail(PCR_HIP_CUDA_ERROR, "moment path: band larger than planned");
        const BinGeom& b = pb.bins;
        total_bins += b.nbins;
        const int64_t cells = (int64_t)g.W * g.st_rows;
        float* mom_v = (mask & 1) ? d_mom : nullptr;
        float* mom_w = (mask & 2) ? d_mom + ((mask & 1) ? (int64_t)p.P * cells : 0) : nullptr;
        const int64_t win_off = (int64_t)mb.win_r0 * g.W;             // the window's first row in the state planes

        PCR_HIP_TRY(hipMemsetAsync(d_fbc, 0, 4, e->stream));           // this band's list (painted at the end of the band)
        b16::Buffers bb{};
        rc = b16::bin(e, g, b, MomentMaker{}, x, y, v, n, (unsigned)kSortChunk, L, &bb);
        if (rc) return rc;
        {
            ScopedKernelTimer t(e, "k_tile_moments");
            if (mask == 1) dispatch_moments<1>(e, g, pb, bb.records, bb.bin_start, mom_v, mom_w, cells);
            else if (mask == 2) dispatch_moments<2>(e, g, pb, bb.records, bb.bin_start, mom_v, mom_w, cells);
            else dispatch_moments<3>(e, g, pb, bb.records, bb.bin_start, mom_v, mom_w, cells);
        }
        // convolutions, per plane kind
        const dim3 col_grid((g.W + 63) / 64, g.tiles_y * yblocks, p.K + 1);
        auto launch_col_mfma = [&](auto kernel, const float* src) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_lds_mfma);
            hipLaunchKernelGGL(kernel, dim3(col_grid.x, col_grid.y, 1), dim3(256), col_lds_mfma, e->stream, g, p.K, p.r, yblocks, taps_y, src, cells, d_u);
        };
        auto launch_col_mfma8 = [&](auto kernel, const float* src) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_lds_mfma8);
            hipLaunchKernelGGL(kernel, dim3(col_grid.x, g.tiles_y * yblocks8, 1), dim3(512), col_lds_mfma8, e->stream, g, p.K, p.r, yblocks8, taps_y, src, cells, d_u);
        };
        auto launch_col = [&](auto kernel, const float* src) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_lds);
            hipLaunchKernelGGL(kernel, col_grid, dim3(256), col_lds, e->stream, g, p.K, p.r, yblocks, taps_y, src, cells, d_u);
        };
        const dim3 row_grid(g.tiles_x * xunits, ((g.st_rows + rw - 1) / rw + 3) / 4);
        auto launch_row = [&](auto kernel, const float* src, float* outp) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds);
            hipLaunchKernelGGL(kernel, row_grid, dim3(256), row_lds, e->stream, g, p.K, p.r, xunits, rs, taps_x, src, cells, outp,
                               store ? 1 : 0);
        };
        for (int kind = 0; kind < 2; ++kind) {
            const float* mom = kind == 0 ? mom_v : mom_w;
            float* outp = (kind == 0 ? pl.sum : pl.wgt);
            if (!mom) continue;
            outp += win_off;
            {
                ScopedKernelTimer t(e, "k_conv_col");
                // matrix-core sweep from r = 24 (15 % faster at r = 48, 7 % slower at r = 12 where the pass is HBM-bound
                // either way)
                if (p.r >= 24) {
                    const int ni = (col_rows_mfma + 15) / 16;         // 16-row staging rounds (four waves)
                    const int ni8 = (col_rows_mfma8 + 31) / 32;       // 32-row staging rounds (eight waves)
                    const bool vec = g.W % 4 == 0 && cells % 4 == 0 && (reinterpret_cast<uintptr_t>(mom) & 15) == 0;
                    if (vec && col_nw == 8 && ni8 <= 6) launch_col_mfma8(&k_conv_col_mfma<6, 8>, mom);
                    else if (vec && col_nw == 8 && ni8 <= 7) launch_col_mfma8(&k_conv_col_mfma<7, 8>, mom);
                    else if (vec && col_nw == 8 && ni8 <= 8) launch_col_mfma8(&k_conv_col_mfma<8, 8>, mom);
                    else if (vec && ni <= 6) launch_col_mfma(&k_conv_col_mfma<6, 4>, mom);
                    else if (vec && ni <= 8) launch_col_mfma(&k_conv_col_mfma<8, 4>, mom);
                    else if (vec && ni <= 11) launch_col_mfma(&k_conv_col_mfma<11, 4>, mom);
                    else if (vec && ni <= 14) launch_col_mfma(&k_conv_col_mfma<14, 4>, mom);
                    else launch_col_mfma(&k_conv_col_mfma<0, 4>, mom);
                } else if (col_rows_per_wave <= 24) launch_col(&k_conv_col<24>, mom);
                else if (col_rows_per_wave <= 32) launch_col(&k_conv_col<32>, mom);
                else if (col_rows_per_wave <= 48) launch_col(&k_conv_col<48>, mom);
                else launch_col(&k_conv_col<0>, mom);
            }
            {
                ScopedKernelTimer t(e, "k_conv_row_accum");
                if (lpr_shift == 6) {
                    if (npr_need <= 20) launch_row(&k_conv_row_accum<6, 20>, d_u, outp);
                    else launch_row(&k_conv_row_accum<6, 0>, d_u, outp);
                } else if (lpr_shift == 4) {
                    if (npr_need <= 6) launch_row(&k_conv_row_accum<4, 6>, d_u, outp);
                    else launch_row(&k_conv_row_accum<4, 0>, d_u, outp);
                } else {
                    if (npr_need <= 3) launch_row(&k_conv_row_accum<2, 3>, d_u, outp);
                    else launch_row(&k_conv_row_accum<2, 0>, d_u, outp);
                }
            }
        }
        {
            // points the expansion cannot represent: painted directly, on the engine's own grid and planes
            ScopedKernelTimer t(e, "k_gauss_list");
            const int fb_blocks = 64;                               // the list is normally empty; grid-strided
            if (mask == 1) hipLaunchKernelGGL(k_gauss_list<1>, dim3(fb_blocks), dim3(256), 0, e->stream, ge, gl, pl, d_fbl, d_fbc, x, y, v);
            else if (mask == 2) hipLaunchKernelGGL(k_gauss_list<2>, dim3(fb_blocks), dim3(256), 0, e->stream, ge, gl, pl, d_fbl, d_fbc, x, y, v);
            else hipLaunchKernelGGL(k_gauss_list<3>, dim3(fb_blocks), dim3(256), 0, e->stream, ge, gl, pl, d_fbl, d_fbc, x, y, v);
        }
    }
    PCR_HIP_TRY(hipGetLastError());
    e->stats.path = 2;
    e->stats.lds_tile_w = kTileW;
    e->stats.lds_tile_h = kTileH;
    e->stats.lds_apron = p.K;           // reported: expansion order
    e->stats.num_bins = total_bins;
    return PCR_HIP_OK;
}

}  // namespace pcrhip
