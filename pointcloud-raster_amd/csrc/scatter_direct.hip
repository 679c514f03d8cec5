// scatter_direct.hip -- the "direct" scatter path: route + accumulate fused, one pass over the
// points, accumulation by global (memory-side) atomics straight into the HBM state planes.
//
// Always applicable (any grid size, any glyph radius); the binned LDS-tile path
// (scatter_binned.hip) takes over where its tiles fit.  Replaces kernel_assign +
// kernel_accumulate_* + kernel_glyph_* of the reference (src/engine/tile_router_kernels.cu:34-61,
// src/engine/accumulator_kernels.cu:31-133, src/engine/glyph_kernels.cu:345-492) but follows the
// CPU semantics (inclusive bounds + clamp, f64 line end points).
#include "engine.hpp"
#include "glyph_device.hpp"

using namespace pcrhip;

namespace {

constexpr int kBlock = 256;

// Wave-aggregated count of valid points: one atomic per wave.
__device__ __forceinline__ void count_valid(unsigned long long* counter, bool valid) {
    unsigned long long m = __ballot(valid);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(counter, (unsigned long long)__popcll(m));
}

// ---- Point glyph -----------------------------------------------------------------
// One thread per point, grid-strided: x/y/value are read once, coalesced (20 B/point);
// each valid point issues one atomic per requested plane.
template <unsigned MASK>
__global__ void __launch_bounds__(kBlock)
k_point_direct(GridDev g, PlanesDev pl, const double* __restrict__ x, const double* __restrict__ y,
               const float* __restrict__ v, uint64_t n, uint32_t* __restrict__ touched,
               unsigned long long* __restrict__ counters) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    // every lane of a wave runs the same number of iterations (ballots inside)
    const uint64_t n_round = ((n + 63) / 64) * 64;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_round; i += stride) {
        bool valid = false;
        int col = 0, row = 0;
        float val = 0.0f;
        if (i < n) {
            valid = point_kept(g, i) && world_to_cell(g, x[i], y[i], col, row);
            valid = valid && row >= g.own_r0 && row < g.own_r1;
            if (MASK & (PCR_HIP_PLANE_SUM | PCR_HIP_PLANE_MAX | PCR_HIP_PLANE_MIN)) val = v[i];
        }
        if (valid) {
            int64_t cell = (int64_t)(row - g.st_r0) * g.W + col;
            if (MASK & PCR_HIP_PLANE_SUM) atomic_add_f32(pl.sum + cell, val);      // SumOp::combine
            if (MASK & PCR_HIP_PLANE_WGT) atomic_add_f32(pl.wgt + cell, 1.0f);     // CountOp::combine (float count)
            if (MASK & PCR_HIP_PLANE_MAX) atomic_max_f32(pl.mx + cell, val);
            if (MASK & PCR_HIP_PLANE_MIN) atomic_min_f32(pl.mn + cell, val);
            touch_tile(g, touched, row, col);
        }
        count_valid(counters, valid);
    }
}

// ---- Glyph sinks -------------------------------------------------------------------
template <unsigned MASK>
struct GlobalSink {
    const GridDev& g;
    PlanesDev pl;
    __device__ __forceinline__ void add(int row, int col, float vw, float w) {
        int64_t cell = (int64_t)(row - g.st_r0) * g.W + col;
        if (MASK & PCR_HIP_PLANE_SUM) atomic_add_f32(pl.sum + cell, vw);
        if (MASK & PCR_HIP_PLANE_WGT) atomic_add_f32(pl.wgt + cell, w);
    }
};

__device__ __forceinline__ float bcast(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
__device__ __forceinline__ int bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// Gaussian: each wave loads 64 points (one per lane, coalesced), prepares their parameters
// in-register, then paints them one after the other with all 64 lanes on one footprint.
template <unsigned MASK>
__global__ void __launch_bounds__(kBlock)
k_gauss_direct(GridDev g, GlyphDev gl, PlanesDev pl, const double* __restrict__ x,
               const double* __restrict__ y, const float* __restrict__ v, uint64_t n,
               uint32_t* __restrict__ touched, unsigned long long* __restrict__ counters) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * kBlock) >> 6;
    GlobalSink<MASK> sink{g, pl};
    for (uint64_t base = wave * 64; base < n; base += nwaves * 64) {
        uint64_t i = base + lane;
        bool valid = false;
        GaussParams q{};
        if (i < n) {
            PointGeom pg = point_geom(g, x[i], y[i]);
            valid = pg.valid && point_kept(g, i);
            if (valid) {
                q = gauss_params(g, gl, pg, v[i], load_chan(gl, i));
                touch_tile(g, touched, pg.row, pg.col);
            }
        }
        count_valid(counters, valid);
        unsigned long long todo = __ballot(valid);
        while (todo) {
            int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            GaussParams u;
            u.val = bcast(q.val, j); u.sub_cx = bcast(q.sub_cx, j); u.sub_cy = bcast(q.sub_cy, j);
            u.sx = bcast(q.sx, j); u.sy = bcast(q.sy, j);
            u.cos_r = bcast(q.cos_r, j); u.sin_r = bcast(q.sin_r, j);
            u.icx = bcast(q.icx, j); u.icy = bcast(q.icy, j); u.r = bcast(q.r, j);
            u.cx0 = bcast(q.cx0, j); u.cx1 = bcast(q.cx1, j);
            u.cy0 = bcast(q.cy0, j); u.cy1 = bcast(q.cy1, j);
            gauss_splat_wave(u, lane, sink);
        }
    }
}

// Line: one lane per point walks its Bresenham segment.
template <unsigned MASK>
__global__ void __launch_bounds__(kBlock)
k_line_direct(GridDev g, GlyphDev gl, PlanesDev pl, const double* __restrict__ x,
              const double* __restrict__ y, const float* __restrict__ v, uint64_t n,
              uint32_t* __restrict__ touched, unsigned long long* __restrict__ counters) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t n_round = ((n + 63) / 64) * 64;
    GlobalSink<MASK> sink{g, pl};
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_round; i += stride) {
        bool valid = false;
        if (i < n) {
            PointGeom pg = point_geom(g, x[i], y[i]);
            valid = pg.valid && point_kept(g, i);
            if (valid) {
                LineParams q = line_params(g, gl, pg, v[i], load_chan(gl, i));
                touch_tile(g, touched, pg.row, pg.col);
                line_walk(q, sink);
            }
        }
        count_valid(counters, valid);
    }
}

inline int blocks_for(uint64_t n, int per_cu, int num_cus) {
    uint64_t b = (n + kBlock - 1) / kBlock;
    uint64_t cap = (uint64_t)per_cu * num_cus;
    return (int)std::max<uint64_t>(1, std::min<uint64_t>(b, cap));
}

#define PCR_DISPATCH_POINT_MASK(M)                                                              \
    case M:                                                                                     \
        hipLaunchKernelGGL((k_point_direct<M>), dim3(blocks), dim3(kBlock), 0, e->stream, e->gd, \
                           pl, x, y, v, n, e->d_touched, e->d_counters);                        \
        break;

}  // namespace

namespace pcrhip {

int direct_point(pcr_hip_engine* e, uint32_t mask, const PlanesDev& pl,
                 const double* x, const double* y, const float* v, uint64_t n) {
    int blocks = blocks_for(n, 8, e->num_cus);
    ScopedKernelTimer t(e, "k_point_direct");
    switch (mask) {
        PCR_DISPATCH_POINT_MASK(1)  PCR_DISPATCH_POINT_MASK(2)  PCR_DISPATCH_POINT_MASK(3)
        PCR_DISPATCH_POINT_MASK(4)  PCR_DISPATCH_POINT_MASK(5)  PCR_DISPATCH_POINT_MASK(6)
        PCR_DISPATCH_POINT_MASK(7)  PCR_DISPATCH_POINT_MASK(8)  PCR_DISPATCH_POINT_MASK(9)
        PCR_DISPATCH_POINT_MASK(10) PCR_DISPATCH_POINT_MASK(11) PCR_DISPATCH_POINT_MASK(12)
        PCR_DISPATCH_POINT_MASK(13) PCR_DISPATCH_POINT_MASK(14) PCR_DISPATCH_POINT_MASK(15)
        default: return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_point: empty plane mask");
    }
    PCR_HIP_TRY(hipGetLastError());
    e->stats.path = 0;
    return PCR_HIP_OK;
}

int direct_glyph(pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask, const PlanesDev& pl,
                 const double* x, const double* y, const float* v, uint64_t n) {
    if (gl.type == PCR_HIP_GLYPH_GAUSSIAN) {
        // one wave per 64-point batch; keep every CU busy with 8 blocks
        ScopedKernelTimer t(e, "k_gauss_direct");
        uint64_t batches = (n + 63) / 64;
        int blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((batches + 3) / 4, (uint64_t)8 * e->num_cus));
        switch (mask) {
            case 1: hipLaunchKernelGGL((k_gauss_direct<1>), dim3(blocks), dim3(kBlock), 0, e->stream, e->gd, gl, pl, x, y, v, n, e->d_touched, e->d_counters); break;
            case 2: hipLaunchKernelGGL((k_gauss_direct<2>), dim3(blocks), dim3(kBlock), 0, e->stream, e->gd, gl, pl, x, y, v, n, e->d_touched, e->d_counters); break;
            case 3: hipLaunchKernelGGL((k_gauss_direct<3>), dim3(blocks), dim3(kBlock), 0, e->stream, e->gd, gl, pl, x, y, v, n, e->d_touched, e->d_counters); break;
            default: return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_glyph: plane mask must be SUM and/or WGT");
        }
    } else if (gl.type == PCR_HIP_GLYPH_LINE) {
        ScopedKernelTimer t(e, "k_line_direct");
        int blocks = blocks_for(n, 8, e->num_cus);
        switch (mask) {
            case 1: hipLaunchKernelGGL((k_line_direct<1>), dim3(blocks), dim3(kBlock), 0, e->stream, e->gd, gl, pl, x, y, v, n, e->d_touched, e->d_counters); break;
            case 2: hipLaunchKernelGGL((k_line_direct<2>), dim3(blocks), dim3(kBlock), 0, e->stream, e->gd, gl, pl, x, y, v, n, e->d_touched, e->d_counters); break;
            case 3: hipLaunchKernelGGL((k_line_direct<3>), dim3(blocks), dim3(kBlock), 0, e->stream, e->gd, gl, pl, x, y, v, n, e->d_touched, e->d_counters); break;
            default: return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_glyph: plane mask must be SUM and/or WGT");
        }
    } else {
        return fail(PCR_HIP_NOT_IMPLEMENTED, "glyph: unknown glyph type");
    }
    PCR_HIP_TRY(hipGetLastError());
    e->stats.path = 0;
    return PCR_HIP_OK;
}

}  // namespace pcrhip
