// common.hpp -- shared host/device definitions of libpcr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>

#include "pcr_hip.h"

namespace pcrhip {

// ---- error plumbing ---------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define PCR_HIP_TRY(expr)                                                              \
    do {                                                                               \
        hipError_t pcr_e_ = (expr);                                                    \
        if (pcr_e_ != hipSuccess)                                                      \
            return ::pcrhip::fail(                                                     \
                pcr_e_ == hipErrorOutOfMemory ? PCR_HIP_OUT_OF_MEMORY : PCR_HIP_CUDA_ERROR, \
                std::string("HIP error: ") + hipGetErrorString(pcr_e_) + " (" #expr ")"); \
    } while (0)

#define PCR_REQUIRE(cond, msg)                                                         \
    do {                                                                               \
        if (!(cond)) return ::pcrhip::fail(PCR_HIP_INVALID_ARGUMENT, msg);             \
    } while (0)

// ---- device-side grid descriptor --------------------------------------------
struct GridDev {
    double min_x, min_y, max_x, max_y;
    double csx, csy;
    double inv_csx, inv_csy;          // 1.0 / cell size, as the glyph code uses (glyph_kernels.cu:97-98)
    int W, H;
    int tw, th, tiles_x, tiles_y;
    int own_r0, own_r1;
    int st_r0, st_rows;
    const unsigned char* mask;        // optional point filter: mask[i] == 0 -> point i is ignored
};

struct PlanesDev {
    float* sum;
    float* wgt;
    float* mx;
    float* mn;
};

// Several reductions over one group's planes in one sweep (k_finalize_group; the Point tile pass when it also finalizes).
struct FinalizeOuts {
    int n;
    int rtype[PCR_HIP_MAX_FINALIZE_OUTPUTS];
    float* out[PCR_HIP_MAX_FINALIZE_OUTPUTS];
};

__device__ __forceinline__ float finalize_rt(int rt, float s, float w, float mx, float mn) {
    switch (rt) {
        case PCR_HIP_SUM: return s;
        case PCR_HIP_COUNT: return w > 0.0f ? w : NAN;
        case PCR_HIP_MAX: return mx == -FLT_MAX ? NAN : mx;
        case PCR_HIP_MIN: return mn == FLT_MAX ? NAN : mn;
        default: return w > 0.0f ? s / w : NAN;
    }
}

struct GlyphDev {
    int type;
    float def_direction, def_half_length, def_sigma_x, def_sigma_y, def_rotation, max_radius;
    const float* direction;
    const float* half_length;
    const float* sigma_x;
    const float* sigma_y;
    const float* rotation;
};

inline GridDev make_grid_dev(const pcr_hip_grid& g) {
    GridDev d;
    d.min_x = g.min_x; d.min_y = g.min_y; d.max_x = g.max_x; d.max_y = g.max_y;
    d.csx = g.cell_size_x; d.csy = g.cell_size_y;
    d.inv_csx = 1.0 / g.cell_size_x; d.inv_csy = 1.0 / g.cell_size_y;
    d.W = g.width; d.H = g.height;
    d.tw = g.tile_width; d.th = g.tile_height;
    d.tiles_x = (g.width + g.tile_width - 1) / g.tile_width;
    d.tiles_y = (g.height + g.tile_height - 1) / g.tile_height;
    d.own_r0 = g.own_row0; d.own_r1 = g.own_row1;
    d.st_r0 = g.state_row0; d.st_rows = g.state_rows;
    d.mask = nullptr;
    return d;
}

int validate_grid(const pcr_hip_grid* g);

#if defined(__HIPCC__)

// n / d for 0 <= n, d >= 1 without the ~25-instruction integer division sequence: one float multiply plus a
// correction step, exact for n < 2^24 and d >= 8 (float holds n exactly and the product is off by less than
// 2^-3 * n / 2^24 < 1); anything else divides for real.  The per-point routing does four of these (tile and bin
// of a cell): they were most of k_bin_count's VALU work.
__device__ __forceinline__ int fast_div(int n, int d) {
    if (n >= (1 << 24) || d < 8) return n / d;
    int q = (int)((float)n * __frcp_rn((float)d));
    int r = n - q * d;
    q += (r >= d) - (r < 0);
    return q;
}

// ---- routing: GridConfig::world_to_cell (src/core/grid_config.cpp:24-43) ------
// Inclusive bounds (BBox::contains, src/core/types.cpp:41-43), floor of a TRUE f64
// division, clamp.  NaN coordinates fail the bounds test, as on the CPU.
// floor(fl(a / cs)) without paying an IEEE f64 division per point: a * (1/cs) is within
// 3.4e-16*|q| of the correctly rounded quotient, so the two can only floor differently when the
// product sits within a few ulps of an integer -- only those lanes take the true division.
__device__ __forceinline__ double floor_quotient(double a, double cs, double inv_cs) {
    double q = a * inv_cs;
    double fl = floor(q);
    double frac = q - fl;
    double eps = fmax(fabs(q), 1.0) * 8.9e-16;
    if (!(frac >= eps && frac <= 1.0 - eps)) fl = floor(a / cs);     // also catches NaN/inf
    return fl;
}

__device__ __forceinline__ bool world_to_cell(const GridDev& g, double wx, double wy, int& col, int& row) {
    if (!(wx >= g.min_x && wx <= g.max_x && wy >= g.min_y && wy <= g.max_y)) return false;
    int c = (int)floor_quotient(wx - g.min_x, g.csx, g.inv_csx);
    int r = (int)floor_quotient(wy - g.max_y, g.csy, g.inv_csy);
    c = max(0, min(c, g.W - 1));
    r = max(0, min(r, g.H - 1));
    col = c;
    row = r;
    return true;
}

// ---- streamed loads --------------------------------------------------------------------------
// Every array of the binned path is read exactly once per pass (x, y, keys, values, records): non-temporal loads keep them
// from displacing what the caches are there for (record lines being filled, the state planes).  Measured in round 4 on the
// Point step, A/B inside one gpurun call: k_bin_count, k_bin_scatter and k_tile_accum each 3-6 % faster, the C2 step
// 0.658 -> 0.642 ms (round 2 had tried the count pass alone and seen nothing); the box's own float4 copy runs at 6.4 TB/s
// with non-temporal accesses and 5.9 without.
typedef double pcr_d2v __attribute__((ext_vector_type(2)));
typedef unsigned pcr_u4v __attribute__((ext_vector_type(4)));
typedef unsigned pcr_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 stream_load(const double2* p) {
    const pcr_d2v t = __builtin_nontemporal_load(reinterpret_cast<const pcr_d2v*>(p));
    return make_double2(t.x, t.y);
}
__device__ __forceinline__ uint4 stream_load(const uint4* p) {
    const pcr_u4v t = __builtin_nontemporal_load(reinterpret_cast<const pcr_u4v*>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ float4 stream_load(const float4* p) {
    typedef float pcr_f4v __attribute__((ext_vector_type(4)));
    const pcr_f4v t = __builtin_nontemporal_load(reinterpret_cast<const pcr_f4v*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint2 stream_load(const uint2* p) {
    const pcr_u2v t = __builtin_nontemporal_load(reinterpret_cast<const pcr_u2v*>(p));
    return make_uint2(t.x, t.y);
}

// A routing kernel's wave-uniform constants all want scalar registers -- eight doubles of grid geometry, the tile and bin
// geometry with the float reciprocals fast_div derives from them, five or six array bases, the loop state -- and the 102 a
// wave has do not hold them: round 4's count passes spilled 21-47 of them to VGPR lanes and read them back inside the point
// loop.  The routing arithmetic is vector work anyway (an f64 VALU instruction takes one scalar operand at most), so the
// geometry is moved into VECTOR registers once, behind an asm the compiler cannot see through (it would move the values back).
// How far each family of routing kernels goes: 0 = everything scalar (round 4), 1 = the eight doubles, 2 = + the integer
// geometry.  A/B'd in one call (profiles/r05_sgpr_spills.md): level 1 ends the spills of the headline's count pass and is
// neutral everywhere; level 2 ends nearly all of them and COSTS -- the vector registers it takes push k_b16_count, the
// Line scatter pass and the multi-tile k_bin_count over an occupancy step (+0.05 to +0.17 ms) -- so level 1 it is: what is
// left spilled is read back with one v_readlane per use and shows in no timing.
#ifndef PCR_VRES_POINT
#define PCR_VRES_POINT 1
#endif
#ifndef PCR_VRES_B16_COUNT
#define PCR_VRES_B16_COUNT 1
#endif
#ifndef PCR_VRES_B16_SCATTER
#define PCR_VRES_B16_SCATTER 1
#endif
__device__ __forceinline__ double vector_resident(double uniform) {
    double v;
    asm volatile("v_mov_b64 %0, %1" : "=v"(v) : "s"(uniform));
    return v;
}
__device__ __forceinline__ int vector_resident(int uniform) {
    int v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(uniform));
    return v;
}
// ints: also the tile geometry (the multi-tile variants, whose touched-tile bookkeeping needs still more scalars)
template <int LEVEL>
__device__ __forceinline__ GridDev vector_resident(const GridDev& u) {
    constexpr bool INTS = LEVEL >= 2;
    GridDev g = u;
    if (LEVEL <= 0) return g;
    g.min_x = vector_resident(u.min_x); g.max_x = vector_resident(u.max_x);
    g.min_y = vector_resident(u.min_y); g.max_y = vector_resident(u.max_y);
    g.csx = vector_resident(u.csx); g.csy = vector_resident(u.csy);
    g.inv_csx = vector_resident(u.inv_csx); g.inv_csy = vector_resident(u.inv_csy);
    if (INTS) {
        g.own_r0 = vector_resident(u.own_r0); g.own_r1 = vector_resident(u.own_r1);
        g.st_r0 = vector_resident(u.st_r0);
        g.W = vector_resident(u.W); g.H = vector_resident(u.H);
        g.tw = vector_resident(u.tw); g.th = vector_resident(u.th); g.tiles_x = vector_resident(u.tiles_x);
    }
    return g;
}

// Point filter (FilterSpec): applied where a kernel decides a point's validity.
__device__ __forceinline__ bool point_kept(const GridDev& g, uint64_t i) { return g.mask == nullptr || g.mask[i] != 0; }

// ---- float atomics ------------------------------------------------------------
// Sum planes: hardware global_atomic_add_f32 / ds_add_f32 (no CAS loop; built with
// -munsafe-fp-atomics).
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// fmaxf / fminf folds (MaxOp/MinOp::combine, include/pcr/ops/builtin_ops.h:27,40) as
// integer atomics on the float's bits: for sign-clear values the signed-int order is the
// float order; for sign-set values the unsigned order is the reverse float order.
// NaN values are skipped (fmaxf(acc, NaN) == acc).  Works for global and LDS addresses.
__device__ __forceinline__ void atomic_max_f32(float* p, float v) {
    if (v != v) return;
    unsigned b = __float_as_uint(v);
    if (!(b & 0x80000000u)) atomicMax(reinterpret_cast<int*>(p), (int)b);
    else atomicMin(reinterpret_cast<unsigned*>(p), b);
}
__device__ __forceinline__ void atomic_min_f32(float* p, float v) {
    if (v != v) return;
    unsigned b = __float_as_uint(v);
    if (!(b & 0x80000000u)) atomicMin(reinterpret_cast<int*>(p), (int)b);
    else atomicMax(reinterpret_cast<unsigned*>(p), b);
}

// Mark the reference tile of (row, col) as having state (pipeline.cpp:688-691, 1220).
__device__ __forceinline__ void touch_tile(const GridDev& g, uint32_t* touched, int row, int col) {
    int t = fast_div(row, g.th) * g.tiles_x + fast_div(col, g.tw);
    // agent-scope relaxed load: served by L2, never by a stale L1 line, so the flag is
    // written only until the first store lands.
    if (__hip_atomic_load(touched + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        __hip_atomic_store(touched + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same for a whole workgroup of a counting pass: on a grid of up to kTouchLdsTiles reference tiles the flags are first
// gathered in LDS (plain stores, every writer stores 1) and only the tiles the workgroup saw are flagged in memory at its end
// -- one L2 round trip per (workgroup, tile) instead of one per POINT (the agent-scope load above is never served by the L1:
// on a C5 shard, four tiles, it was a dependent L2 access for each of the 125 M points of the count pass).
constexpr int kTouchLdsTiles = 1024;
struct TouchLds {
    unsigned* flags;            // [kTouchLdsTiles] in LDS, zeroed by begin()
    bool on;                    // false: one tile (the caller flags it itself) or too many tiles (touch_tile per point)
    __device__ __forceinline__ void begin(const GridDev& g, unsigned* lds, int threads) {
        flags = lds;
        const int nt = g.tiles_x * g.tiles_y;
        on = nt > 1 && nt <= kTouchLdsTiles;
        if (on) for (int i = threadIdx.x; i < nt; i += threads) flags[i] = 0u;
    }
    __device__ __forceinline__ void touch(const GridDev& g, uint32_t* touched, int row, int col) const {
        if (on) flags[fast_div(row, g.th) * g.tiles_x + fast_div(col, g.tw)] = 1u;
        else touch_tile(g, touched, row, col);
    }
    // after a barrier that orders the workgroup's touch() calls
    __device__ __forceinline__ void flush(const GridDev& g, uint32_t* touched, int threads) const {
        if (!on) return;
        const int nt = g.tiles_x * g.tiles_y;
        for (int i = threadIdx.x; i < nt; i += threads)
            if (flags[i] && __hip_atomic_load(touched + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                __hip_atomic_store(touched + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

#endif  // __HIPCC__

}  // namespace pcrhip
