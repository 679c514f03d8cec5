// state_kernels.hip -- identity fill / merge / finalize of the band-sequential state planes.
// Replaces the reference's init_state_kernel / merge_state_kernel / finalize_kernel
// (src/engine/grid_merge.cu:16-45) and the CPU finalize + band assembly of
// src/engine/pipeline.cpp:1204-1286.  All HBM-streaming, one pass, float4 wide.
#include "common.hpp"

#include <cstdlib>

using namespace pcrhip;

namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t work_items) {
    int64_t blocks = (work_items + kBlock - 1) / kBlock;
    // 256 CUs x 8 resident blocks; the rest is grid-strided.
    return (int)std::max<int64_t>(1, std::min<int64_t>(blocks, 2048));
}

__global__ void __launch_bounds__(kBlock) k_fill(float* __restrict__ p, float v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t n4 = n >> 2;
    float4 v4 = make_float4(v, v, v, v);
    float4* p4 = reinterpret_cast<float4*>(p);
    for (int64_t j = i; j < n4; j += stride) p4[j] = v4;
    for (int64_t j = (n4 << 2) + i; j < n; j += stride) p[j] = v;
}

// kind: 0 = add, 1 = fmaxf, 2 = fminf  (Op::merge, include/pcr/ops/builtin_ops.h:15,28,41,54,67,95-97)
template <int KIND>
__device__ __forceinline__ float merge1(float a, float b) {
    if (KIND == 0) return a + b;
    if (KIND == 1) return fmaxf(a, b);
    return fminf(a, b);
}

template <int KIND>
__global__ void __launch_bounds__(kBlock) k_merge(float* __restrict__ d, const float* __restrict__ s, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t n4 = n >> 2;
    float4* d4 = reinterpret_cast<float4*>(d);
    const float4* s4 = reinterpret_cast<const float4*>(s);
    for (int64_t j = i; j < n4; j += stride) {
        float4 a = d4[j], b = s4[j];
        a.x = merge1<KIND>(a.x, b.x);
        a.y = merge1<KIND>(a.y, b.y);
        a.z = merge1<KIND>(a.z, b.z);
        a.w = merge1<KIND>(a.w, b.w);
        d4[j] = a;
    }
    for (int64_t j = (n4 << 2) + i; j < n; j += stride) d[j] = merge1<KIND>(d[j], s[j]);
}

// Op::finalize (builtin_ops.h:16,29,42,55,68-70,99-101): Sum -> sum (0.0 when empty),
// Count -> count>0 ? count : NaN, Average/WeightedAverage -> den>0 ? num/den : NaN,
// Max/Min -> identity ? NaN : value.
template <int RT>
__device__ __forceinline__ float finalize1(float a, float b) {
    if (RT == PCR_HIP_SUM) return a;
    if (RT == PCR_HIP_COUNT) return a > 0.0f ? a : NAN;
    if (RT == PCR_HIP_MAX) return a == -FLT_MAX ? NAN : a;
    if (RT == PCR_HIP_MIN) return a == FLT_MAX ? NAN : a;
    return b > 0.0f ? a / b : NAN;     // Average / WeightedAverage: a = numerator, b = denominator
}

// One thread = 4 consecutive cells of one row when W % 4 == 0, else 1 cell.
template <int RT, int VEC>
__global__ void __launch_bounds__(kBlock)
k_finalize(GridDev g, const float* __restrict__ pa, const float* __restrict__ pb,
           const uint32_t* __restrict__ touched, float* __restrict__ out) {
    const int rows = g.own_r1 - g.own_r0;
    const int64_t items = (int64_t)rows * (g.W / VEC);
    const int per_row = g.W / VEC;
    const bool one_tile = (g.tiles_x * g.tiles_y == 1);
    // untouched tile -> NaN (band pre-filled with NaN and tiles without state skipped,
    // pipeline.cpp:1204-1222); with one tile that is a single wave-uniform flag
    const bool all_touched = (touched == nullptr) || (one_tile && touched[0] != 0u);
    const bool none_touched = (touched != nullptr) && one_tile && touched[0] == 0u;
    for (int64_t it = (int64_t)blockIdx.x * kBlock + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * kBlock) {
        int r = (int)(it / per_row);
        int c = (int)(it - (int64_t)r * per_row) * VEC;
        int row = g.own_r0 + r;
        int64_t si = (int64_t)(row - g.st_r0) * g.W + c;
        int64_t oi = (int64_t)r * g.W + c;
        if (VEC == 4) {
            float4 a = *reinterpret_cast<const float4*>(pa + si);
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (RT == PCR_HIP_AVERAGE || RT == PCR_HIP_WEIGHTED_AVERAGE)
                b = *reinterpret_cast<const float4*>(pb + si);
            float4 o;
            o.x = finalize1<RT>(a.x, b.x);
            o.y = finalize1<RT>(a.y, b.y);
            o.z = finalize1<RT>(a.z, b.z);
            o.w = finalize1<RT>(a.w, b.w);
            if (none_touched) {
                o = make_float4(NAN, NAN, NAN, NAN);
            } else if (!all_touched) {
                int trow = (row / g.th) * g.tiles_x;
                if (!touched[trow + (c + 0) / g.tw]) o.x = NAN;
                if (!touched[trow + (c + 1) / g.tw]) o.y = NAN;
                if (!touched[trow + (c + 2) / g.tw]) o.z = NAN;
                if (!touched[trow + (c + 3) / g.tw]) o.w = NAN;
            }
            *reinterpret_cast<float4*>(out + oi) = o;
        } else {
            float a = pa[si];
            float b = (RT == PCR_HIP_AVERAGE || RT == PCR_HIP_WEIGHTED_AVERAGE) ? pb[si] : 0.f;
            float o = finalize1<RT>(a, b);
            if (none_touched) o = NAN;
            else if (!all_touched && !touched[(row / g.th) * g.tiles_x + c / g.tw]) o = NAN;
            out[oi] = o;
        }
    }
}

// Several reductions over one group's planes in one sweep (FinalizeOuts, finalize_rt: common.hpp).

template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_finalize_group(GridDev g, PlanesDev pl, unsigned need, const uint32_t* __restrict__ touched, FinalizeOuts fo,
                 const uint32_t* __restrict__ bands_done) {
    if (bands_done && *bands_done != 0u) return;              // the scatter that defined the planes stored the bands as well
    const int rows = g.own_r1 - g.own_r0;
    const int per_row = g.W / VEC;
    const int64_t items = (int64_t)rows * per_row;
    const bool one_tile = (g.tiles_x * g.tiles_y == 1);
    const bool all_touched = (touched == nullptr) || (one_tile && touched[0] != 0u);
    const bool none_touched = (touched != nullptr) && one_tile && touched[0] == 0u;
    for (int64_t it = (int64_t)blockIdx.x * kBlock + threadIdx.x; it < items; it += (int64_t)gridDim.x * kBlock) {
        int r = (int)(it / per_row);
        int c = (int)(it - (int64_t)r * per_row) * VEC;
        int row = g.own_r0 + r;
        int64_t si = (int64_t)(row - g.st_r0) * g.W + c;
        int64_t oi = (int64_t)r * g.W + c;
        float s[VEC], w[VEC], mx[VEC], mn[VEC];
        bool live[VEC];
        if (VEC == 4) {
            float4 t;
            if (need & 1) { t = stream_load(reinterpret_cast<const float4*>(pl.sum + si)); s[0] = t.x; s[1] = t.y; s[2] = t.z; s[3] = t.w; }
            if (need & 2) { t = stream_load(reinterpret_cast<const float4*>(pl.wgt + si)); w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w; }
            if (need & 4) { t = stream_load(reinterpret_cast<const float4*>(pl.mx + si)); mx[0] = t.x; mx[1] = t.y; mx[2] = t.z; mx[3] = t.w; }
            if (need & 8) { t = stream_load(reinterpret_cast<const float4*>(pl.mn + si)); mn[0] = t.x; mn[1] = t.y; mn[2] = t.z; mn[3] = t.w; }
        } else {
            if (need & 1) s[0] = pl.sum[si];
            if (need & 2) w[0] = pl.wgt[si];
            if (need & 4) mx[0] = pl.mx[si];
            if (need & 8) mn[0] = pl.mn[si];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if (!(need & 1)) s[k] = 0.f;
            if (!(need & 2)) w[k] = 0.f;
            if (!(need & 4)) mx[k] = -FLT_MAX;
            if (!(need & 8)) mn[k] = FLT_MAX;
            live[k] = all_touched || (!none_touched && touched[(row / g.th) * g.tiles_x + (c + k) / g.tw] != 0u);
        }
        for (int o = 0; o < fo.n; ++o) {
            float v[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] = live[k] ? finalize_rt(fo.rtype[o], s[k], w[k], mx[k], mn[k]) : NAN;
            if (VEC == 4) {
                // finished bands: written once, read by nobody on the device (C2 step -1.5 % against plain stores)
                typedef float f4v __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(f4v{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4v*>(fo.out[o] + oi));
            } else fo.out[o][oi] = v[0];
        }
    }
}

// FilterSpec evaluation: AND of predicates (evaluate_predicate, src/engine/filter.cpp:34-56).
struct PredSet {
    int n;
    pcr_hip_predicate p[PCR_HIP_MAX_FILTER_PREDICATES];
};

__device__ __forceinline__ bool eval_pred(const pcr_hip_predicate& pr, float v) {
    switch (pr.op) {
        case PCR_HIP_CMP_EQUAL: return v == pr.value;
        case PCR_HIP_CMP_NOT_EQUAL: return v != pr.value;
        case PCR_HIP_CMP_LESS: return v < pr.value;
        case PCR_HIP_CMP_LESS_EQUAL: return v <= pr.value;
        case PCR_HIP_CMP_GREATER: return v > pr.value;
        case PCR_HIP_CMP_GREATER_EQUAL: return v >= pr.value;
        case PCR_HIP_CMP_IN_SET: {
            bool in = false;
            for (int k = 0; k < pr.set_size; ++k) in = in || (v == pr.set[k]);
            return in;
        }
        case PCR_HIP_CMP_NOT_IN_SET: {
            bool in = false;
            for (int k = 0; k < pr.set_size; ++k) in = in || (v == pr.set[k]);
            return !in;
        }
    }
    return false;
}

__global__ void __launch_bounds__(kBlock)
k_filter_mask(PredSet ps, uint64_t n, unsigned char* __restrict__ mask, unsigned long long* __restrict__ pass_count) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t n_round = ((n + 63) / 64) * 64;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_round; i += stride) {
        bool keep = i < n;
        if (keep)
            for (int k = 0; k < ps.n; ++k) keep = keep && eval_pred(ps.p[k], ps.p[k].d_channel[i]);
        if (i < n) mask[i] = keep ? 1 : 0;
        unsigned long long m = __ballot(keep);
        if (pass_count && (threadIdx.x & 63) == 0 && m) atomicAdd(pass_count, (unsigned long long)__popcll(m));
    }
}

template <int RT>
int launch_finalize(const GridDev& g, const float* pa, const float* pb, const uint32_t* touched,
                    float* out, hipStream_t s) {
    const int rows = g.own_r1 - g.own_r0;
    if (rows <= 0) return PCR_HIP_OK;
    bool vec = (g.W % 4 == 0) && ((reinterpret_cast<uintptr_t>(pa) & 15) == 0) &&
               ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
               (pb == nullptr || (reinterpret_cast<uintptr_t>(pb) & 15) == 0);
    if (vec) {
        int64_t items = (int64_t)rows * (g.W / 4);
        hipLaunchKernelGGL((k_finalize<RT, 4>), dim3(grid_for(items)), dim3(kBlock), 0, s, g, pa, pb, touched, out);
    } else {
        int64_t items = (int64_t)rows * g.W;
        hipLaunchKernelGGL((k_finalize<RT, 1>), dim3(grid_for(items)), dim3(kBlock), 0, s, g, pa, pb, touched, out);
    }
    PCR_HIP_TRY(hipGetLastError());
    return PCR_HIP_OK;
}

int merge_kind(int kind, float* d, const float* s, int64_t n, hipStream_t st) {
    if (n <= 0) return PCR_HIP_OK;
    PCR_REQUIRE(d && s, "merge: null plane");
    PCR_REQUIRE(((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0,
                "merge: planes must be 16-byte aligned");
    dim3 gr(grid_for((n + 3) / 4)), bl(kBlock);
    if (kind == 0) hipLaunchKernelGGL(k_merge<0>, gr, bl, 0, st, d, s, n);
    else if (kind == 1) hipLaunchKernelGGL(k_merge<1>, gr, bl, 0, st, d, s, n);
    else hipLaunchKernelGGL(k_merge<2>, gr, bl, 0, st, d, s, n);
    PCR_HIP_TRY(hipGetLastError());
    return PCR_HIP_OK;
}

}  // namespace

extern "C" {

int pcr_hip_state_floats(int rtype, int* k) {
    PCR_REQUIRE(k, "state_floats: null out pointer");
    switch (rtype) {
        case PCR_HIP_SUM: case PCR_HIP_MAX: case PCR_HIP_MIN: case PCR_HIP_COUNT: *k = 1; return PCR_HIP_OK;
        case PCR_HIP_AVERAGE: case PCR_HIP_WEIGHTED_AVERAGE: *k = 2; return PCR_HIP_OK;
        default: *k = 0; return fail(PCR_HIP_INVALID_ARGUMENT, "pipeline: unknown reduction type");
    }
}

int pcr_hip_plane_fill(float* d_plane, float value, int64_t cells, pcr_hip_stream s) {
    if (cells <= 0) return PCR_HIP_OK;
    PCR_REQUIRE(d_plane, "plane_fill: null plane");
    PCR_REQUIRE((reinterpret_cast<uintptr_t>(d_plane) & 15) == 0, "plane_fill: plane must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(s);
    if (value == 0.0f && !std::signbit(value)) {
        PCR_HIP_TRY(hipMemsetAsync(d_plane, 0, (size_t)cells * sizeof(float), st));
        return PCR_HIP_OK;
    }
    hipLaunchKernelGGL(k_fill, dim3(grid_for((cells + 3) / 4)), dim3(kBlock), 0, st, d_plane, value, cells);
    PCR_HIP_TRY(hipGetLastError());
    return PCR_HIP_OK;
}

// Op::identity (builtin_ops.h:13,26,39,52,65,82) over K planes of `cells` floats.
int pcr_hip_state_init(int rtype, float* d_state, int64_t cells, pcr_hip_stream s) {
    int k = 0;
    int rc = pcr_hip_state_floats(rtype, &k);
    if (rc) return rc;
    float id = rtype == PCR_HIP_MAX ? -FLT_MAX : rtype == PCR_HIP_MIN ? FLT_MAX : 0.0f;
    return pcr_hip_plane_fill(d_state, id, (int64_t)k * cells, s);
}

int pcr_hip_state_merge(int rtype, float* d_dst, const float* d_src, int64_t cells, pcr_hip_stream s) {
    int k = 0;
    int rc = pcr_hip_state_floats(rtype, &k);
    if (rc) return rc;
    int kind = rtype == PCR_HIP_MAX ? 1 : rtype == PCR_HIP_MIN ? 2 : 0;
    return merge_kind(kind, d_dst, d_src, (int64_t)k * cells, static_cast<hipStream_t>(s));
}

int pcr_hip_plane_merge(uint32_t plane_kind, float* d_dst, const float* d_src, int64_t cells, pcr_hip_stream s) {
    int kind;
    switch (plane_kind) {
        case PCR_HIP_PLANE_SUM: case PCR_HIP_PLANE_WGT: kind = 0; break;
        case PCR_HIP_PLANE_MAX: kind = 1; break;
        case PCR_HIP_PLANE_MIN: kind = 2; break;
        default: return fail(PCR_HIP_INVALID_ARGUMENT, "plane_merge: exactly one PCR_HIP_PLANE_* kind expected");
    }
    return merge_kind(kind, d_dst, d_src, cells, static_cast<hipStream_t>(s));
}

int pcr_hip_filter_mask(const pcr_hip_predicate* preds, int n_pred, uint64_t n, uint8_t* d_mask,
                        unsigned long long* d_pass_count, pcr_hip_stream s) {
    PCR_REQUIRE(n_pred >= 0 && n_pred <= PCR_HIP_MAX_FILTER_PREDICATES, "filter_points: too many predicates (max 16)");
    PCR_REQUIRE(n_pred == 0 || preds, "filter_points: null predicate list");
    hipStream_t st = static_cast<hipStream_t>(s);
    if (d_pass_count) PCR_HIP_TRY(hipMemsetAsync(d_pass_count, 0, sizeof(unsigned long long), st));
    if (n == 0) return PCR_HIP_OK;
    PCR_REQUIRE(d_mask, "filter_points: null mask");
    PredSet ps;
    ps.n = n_pred;
    for (int k = 0; k < n_pred; ++k) {
        PCR_REQUIRE(preds[k].d_channel, "filter_points: null channel pointer");
        PCR_REQUIRE(preds[k].op >= 0 && preds[k].op <= PCR_HIP_CMP_NOT_IN_SET, "filter_points: unknown compare op");
        PCR_REQUIRE(preds[k].set_size >= 0 && preds[k].set_size <= PCR_HIP_MAX_FILTER_SET,
                    "filter_points: value_set larger than 16 entries");
        ps.p[k] = preds[k];
    }
    hipLaunchKernelGGL(k_filter_mask, dim3(grid_for((int64_t)n)), dim3(kBlock), 0, st, ps, n, d_mask, d_pass_count);
    PCR_HIP_TRY(hipGetLastError());
    return PCR_HIP_OK;
}

namespace {
// [own_lo, own_hi): the flags (row-major over the tile grid) whose tiles hold rows this device OWNS.  A flag another rank set
// for a tile this device owns no row of changes none of this device's bands: only a change inside the range drops them.
__global__ void __launch_bounds__(256)
k_touched_union(uint32_t* __restrict__ local, const uint32_t* __restrict__ other, int n, int own_lo, int own_hi,
                uint32_t* __restrict__ done, int n_words) {
    __shared__ unsigned changed;
    if (threadIdx.x == 0) changed = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256)
        if (other[i] != 0u && local[i] == 0u) {
            local[i] = 1u;                                                         // (every writer stores 1)
            if (i >= own_lo && i < own_hi) changed = 1u;
        }
    __syncthreads();
    if (changed && done)
        for (int w = threadIdx.x; w < n_words; w += 256) done[w] = 0u;
}
}  // namespace

int pcr_hip_touched_union_owned(uint32_t* d_local, const uint32_t* d_union, int32_t tiles_x, int32_t tiles_y,
                                int32_t own_tile_row0, int32_t own_tile_row1, uint32_t* d_bands_done, int32_t n_words,
                                pcr_hip_stream s) {
    PCR_REQUIRE(d_local && d_union && tiles_x >= 0 && tiles_y >= 0 && n_words >= 0, "touched_union: bad argument");
    PCR_REQUIRE(own_tile_row0 >= 0 && own_tile_row0 <= own_tile_row1 && own_tile_row1 <= tiles_y,
                "touched_union: owned tile rows outside the tile grid");
    if ((int64_t)tiles_x * tiles_y == 0) return PCR_HIP_OK;
    PCR_REQUIRE((int64_t)tiles_x * tiles_y < (1ll << 31), "touched_union: tile grid too large");
    hipLaunchKernelGGL(k_touched_union, dim3(1), dim3(256), 0, static_cast<hipStream_t>(s), d_local, d_union, tiles_x * tiles_y,
                       own_tile_row0 * tiles_x, own_tile_row1 * tiles_x, d_bands_done, (int)n_words);
    PCR_HIP_TRY(hipGetLastError());
    return PCR_HIP_OK;
}

int pcr_hip_touched_union(uint32_t* d_local, const uint32_t* d_union, int32_t n, uint32_t* d_bands_done, int32_t n_words,
                          pcr_hip_stream s) {
    PCR_REQUIRE(d_local && d_union && n >= 0 && n_words >= 0, "touched_union: bad argument");
    if (n == 0) return PCR_HIP_OK;
    hipLaunchKernelGGL(k_touched_union, dim3(1), dim3(256), 0, static_cast<hipStream_t>(s), d_local, d_union, (int)n, 0, (int)n,
                       d_bands_done, (int)n_words);
    PCR_HIP_TRY(hipGetLastError());
    return PCR_HIP_OK;
}

int pcr_hip_finalize_group(const pcr_hip_grid* g, const pcr_hip_planes* planes, const uint32_t* d_tile_touched,
                           int n_out, const int* rtypes, float* const* d_outs, pcr_hip_stream s) {
    return pcr_hip_finalize_group_unless(g, planes, d_tile_touched, n_out, rtypes, d_outs, nullptr, s);
}

int pcr_hip_finalize_group_unless(const pcr_hip_grid* g, const pcr_hip_planes* planes, const uint32_t* d_tile_touched,
                                  int n_out, const int* rtypes, float* const* d_outs, const uint32_t* d_bands_done,
                                  pcr_hip_stream s) {
    int rc = validate_grid(g);
    if (rc) return rc;
    PCR_REQUIRE(planes && rtypes && d_outs, "finalize_group: null argument");
    PCR_REQUIRE(n_out >= 1 && n_out <= PCR_HIP_MAX_FINALIZE_OUTPUTS, "finalize_group: 1..8 outputs");
    GridDev gd = make_grid_dev(*g);
    const int rows = gd.own_r1 - gd.own_r0;
    if (rows <= 0) return PCR_HIP_OK;
    FinalizeOuts fo;
    fo.n = n_out;
    unsigned need = 0;
    bool aligned = gd.W % 4 == 0;
    for (int i = 0; i < n_out; ++i) {
        fo.rtype[i] = rtypes[i];
        fo.out[i] = d_outs[i];
        PCR_REQUIRE(d_outs[i], "finalize_group: null output band");
        aligned = aligned && (reinterpret_cast<uintptr_t>(d_outs[i]) & 15) == 0;
        switch (rtypes[i]) {
            case PCR_HIP_SUM: need |= 1; break;
            case PCR_HIP_COUNT: need |= 2; break;
            case PCR_HIP_MAX: need |= 4; break;
            case PCR_HIP_MIN: need |= 8; break;
            case PCR_HIP_AVERAGE: case PCR_HIP_WEIGHTED_AVERAGE: need |= 3; break;
            default: return fail(PCR_HIP_INVALID_ARGUMENT, "pipeline: unknown reduction type");
        }
    }
    PCR_REQUIRE(!(need & 1) || planes->d_sum, "finalize_group: sum plane missing");
    PCR_REQUIRE(!(need & 2) || planes->d_wgt, "finalize_group: weight plane missing");
    PCR_REQUIRE(!(need & 4) || planes->d_max, "finalize_group: max plane missing");
    PCR_REQUIRE(!(need & 8) || planes->d_min, "finalize_group: min plane missing");
    PlanesDev pl{planes->d_sum, planes->d_wgt, planes->d_max, planes->d_min};
    for (float* p : {pl.sum, pl.wgt, pl.mx, pl.mn}) aligned = aligned && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    hipStream_t st = static_cast<hipStream_t>(s);
    if (aligned) {
        hipLaunchKernelGGL(k_finalize_group<4>, dim3(grid_for((int64_t)rows * (gd.W / 4))), dim3(kBlock), 0, st,
                           gd, pl, need, d_tile_touched, fo, d_bands_done);
    } else {
        hipLaunchKernelGGL(k_finalize_group<1>, dim3(grid_for((int64_t)rows * gd.W)), dim3(kBlock), 0, st,
                           gd, pl, need, d_tile_touched, fo, d_bands_done);
    }
    PCR_HIP_TRY(hipGetLastError());
    return PCR_HIP_OK;
}

int pcr_hip_finalize(int rtype, const pcr_hip_grid* g, const pcr_hip_planes* planes,
                     const uint32_t* d_tile_touched, float* d_out, pcr_hip_stream s) {
    int rc = validate_grid(g);
    if (rc) return rc;
    PCR_REQUIRE(planes && d_out, "finalize: null argument");
    GridDev gd = make_grid_dev(*g);
    hipStream_t st = static_cast<hipStream_t>(s);
    switch (rtype) {
        case PCR_HIP_SUM:
            PCR_REQUIRE(planes->d_sum, "finalize(Sum): sum plane missing");
            return launch_finalize<PCR_HIP_SUM>(gd, planes->d_sum, nullptr, d_tile_touched, d_out, st);
        case PCR_HIP_COUNT:
            PCR_REQUIRE(planes->d_wgt, "finalize(Count): weight plane missing");
            return launch_finalize<PCR_HIP_COUNT>(gd, planes->d_wgt, nullptr, d_tile_touched, d_out, st);
        case PCR_HIP_MAX:
            PCR_REQUIRE(planes->d_max, "finalize(Max): max plane missing");
            return launch_finalize<PCR_HIP_MAX>(gd, planes->d_max, nullptr, d_tile_touched, d_out, st);
        case PCR_HIP_MIN:
            PCR_REQUIRE(planes->d_min, "finalize(Min): min plane missing");
            return launch_finalize<PCR_HIP_MIN>(gd, planes->d_min, nullptr, d_tile_touched, d_out, st);
        case PCR_HIP_AVERAGE:
        case PCR_HIP_WEIGHTED_AVERAGE:
            PCR_REQUIRE(planes->d_sum && planes->d_wgt, "finalize(Average): sum and weight planes required");
            return launch_finalize<PCR_HIP_AVERAGE>(gd, planes->d_sum, planes->d_wgt, d_tile_touched, d_out, st);
        default:
            return fail(PCR_HIP_INVALID_ARGUMENT, "pipeline: unknown reduction type");
    }
}

}  // extern "C"
