// route.hip -- routing an UNPARTITIONED cloud to the owners of row blocks (multi-GPU ingest).
//
// The reference is single-device: its 1 B-point protocol feeds one pipeline
// (scripts/benchmarks/benchmark_billion_points.py:221-345).  With the grid row-block sharded over
// the GPUs of a node, a rank that is handed an arbitrary shard of the cloud first groups its points
// by owner ("points routed by y"), then the groups travel to their owners point-to-point over xGMI
// (pcr/distributed.py: all_to_all_single on these buffers) and every rank ingests only points it owns.
//
//   pcr_hip_route_count    x, y -> owner byte per point (world_to_cell row -> block), per-owner counts
//   pcr_hip_route_scatter  owner bytes -> every array of the cloud regrouped by owner (counting sort)
//
// The owner decision is GridConfig::world_to_cell (src/core/grid_config.cpp:24-43) exactly as the
// scatter kernels make it (common.hpp), so a routed run drops and keeps the same points as an unrouted one.
#include "common.hpp"

using namespace pcrhip;

namespace {

constexpr int kThreads = 1024;
constexpr int kPer = 8;                          // points per thread and block pass
constexpr int kMaxParts = PCR_HIP_MAX_ROUTE_PARTS;
constexpr unsigned char kNoOwner = 0xFF;

struct Splits {
    int n;
    int row[kMaxParts + 1];                      // part p owns rows [row[p], row[p+1])
};

struct Arrays {
    int n;
    const void* src[PCR_HIP_MAX_ROUTE_ARRAYS];
    void* dst[PCR_HIP_MAX_ROUTE_ARRAYS];
    int elem[PCR_HIP_MAX_ROUTE_ARRAYS];          // 4 or 8 bytes
};

__global__ void __launch_bounds__(kThreads)
k_route_count(GridDev g, Splits sp, const double* __restrict__ x, const double* __restrict__ y, uint64_t n,
              unsigned char* __restrict__ dest, unsigned long long* __restrict__ counts) {
    __shared__ unsigned hist[kMaxParts];
    __shared__ int rows[kMaxParts + 1];
    if (threadIdx.x < kMaxParts) hist[threadIdx.x] = 0;
    if (threadIdx.x <= sp.n) rows[threadIdx.x] = sp.row[threadIdx.x];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * kThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        int col, row;
        unsigned char d = kNoOwner;
        if (world_to_cell(g, x[i], y[i], col, row) && point_kept(g, i)) {
            int lo = 0, hi = sp.n;               // last part whose first row is <= row
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (rows[mid] <= row) lo = mid; else hi = mid;
            }
            if (row >= rows[lo] && row < rows[lo + 1]) {
                d = (unsigned char)lo;
                atomicAdd(&hist[lo], 1u);
            }
        }
        dest[i] = d;
    }
    __syncthreads();
    if (threadIdx.x < sp.n && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

__global__ void __launch_bounds__(kThreads)
k_route_scatter(const unsigned char* __restrict__ dest, uint64_t n, int nparts,
                unsigned long long* __restrict__ cursors, Arrays a) {
    __shared__ unsigned hist[kMaxParts];
    __shared__ unsigned long long base[kMaxParts];
    if (threadIdx.x < kMaxParts) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t first = (uint64_t)blockIdx.x * (kThreads * kPer);
    unsigned char d[kPer];
    unsigned rank[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const uint64_t i = first + (uint64_t)k * kThreads + threadIdx.x;
        d[k] = i < n ? dest[i] : kNoOwner;
        rank[k] = 0;
        if (d[k] != kNoOwner) rank[k] = atomicAdd(&hist[d[k]], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < nparts && hist[threadIdx.x])
        base[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        if (d[k] == kNoOwner) continue;
        const uint64_t i = first + (uint64_t)k * kThreads + threadIdx.x;
        const uint64_t o = base[d[k]] + rank[k];
        for (int q = 0; q < a.n; ++q) {
            if (a.elem[q] == 8) static_cast<uint64_t*>(a.dst[q])[o] = static_cast<const uint64_t*>(a.src[q])[i];
            else static_cast<uint32_t*>(a.dst[q])[o] = static_cast<const uint32_t*>(a.src[q])[i];
        }
    }
}

__global__ void __launch_bounds__(kThreads)
k_absmax(const float* __restrict__ v, const unsigned char* __restrict__ mask, uint64_t n, unsigned* __restrict__ out_bits) {
    float m = 0.f;
    const uint64_t stride = (uint64_t)gridDim.x * kThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const float a = fabsf(v[i]);
        if (a <= FLT_MAX && (!mask || mask[i])) m = fmaxf(m, a);     // skips NaN, inf and filtered-out points
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    // non-negative floats order like their bit patterns
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out_bits, __float_as_uint(m));
}

// max over the finite entries of +v and of -v (each >= 0), one word each
__global__ void __launch_bounds__(kThreads)
k_signed_max(const float* __restrict__ v, const unsigned char* __restrict__ mask, uint64_t n, unsigned* __restrict__ out_bits) {
    float mp = 0.f, mn = 0.f;
    const uint64_t stride = (uint64_t)gridDim.x * kThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const float a = v[i];
        if (fabsf(a) <= FLT_MAX && (!mask || mask[i])) { mp = fmaxf(mp, a); mn = fmaxf(mn, -a); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mp = fmaxf(mp, __shfl_xor(mp, off, 64));
        mn = fmaxf(mn, __shfl_xor(mn, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (mp > 0.f) atomicMax(out_bits, __float_as_uint(mp));
        if (mn > 0.f) atomicMax(out_bits + 1, __float_as_uint(mn));
    }
}

}  // namespace

extern "C" {

int pcr_hip_signed_max_f32_masked(const float* d_values, const uint8_t* d_mask, uint64_t n, uint32_t* d_scratch_2words,
                                  float* h_max_pos, float* h_max_neg, pcr_hip_stream s) {
    PCR_REQUIRE(h_max_pos && h_max_neg, "signed_max_f32: null result pointer");
    *h_max_pos = *h_max_neg = 0.f;
    if (n == 0) return PCR_HIP_OK;
    PCR_REQUIRE(d_values && d_scratch_2words, "signed_max_f32: null array or scratch words");
    hipStream_t st = static_cast<hipStream_t>(s);
    PCR_HIP_TRY(hipMemsetAsync(d_scratch_2words, 0, 2 * sizeof(unsigned), st));
    const uint64_t want = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_signed_max, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(kThreads), 0, st, d_values, d_mask, n, d_scratch_2words);
    PCR_HIP_TRY(hipGetLastError());
    unsigned bits[2] = {0u, 0u};
    PCR_HIP_TRY(hipMemcpyAsync(bits, d_scratch_2words, sizeof bits, hipMemcpyDeviceToHost, st));
    PCR_HIP_TRY(hipStreamSynchronize(st));
    static_assert(sizeof(float) == sizeof(unsigned), "float bits");
    __builtin_memcpy(h_max_pos, &bits[0], sizeof(float));
    __builtin_memcpy(h_max_neg, &bits[1], sizeof(float));
    return PCR_HIP_OK;
}

int pcr_hip_absmax_f32(const float* d_values, uint64_t n, float* h_result, pcr_hip_stream s) {
    return pcr_hip_absmax_f32_masked(d_values, nullptr, n, nullptr, h_result, s);
}

int pcr_hip_absmax_f32_masked(const float* d_values, const uint8_t* d_mask, uint64_t n, uint32_t* d_scratch_word,
                              float* h_result, pcr_hip_stream s) {
    PCR_REQUIRE(h_result, "absmax_f32: null result pointer");
    *h_result = 0.f;
    if (n == 0) return PCR_HIP_OK;
    PCR_REQUIRE(d_values, "absmax_f32: null array");
    hipStream_t st = static_cast<hipStream_t>(s);
    // a caller that asks on every ingest brings its own device word: hipMalloc + hipFree are device-wide syncs
    unsigned* d_bits = d_scratch_word;
    if (!d_bits) PCR_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_bits), sizeof(unsigned)));
    hipError_t err = hipMemsetAsync(d_bits, 0, sizeof(unsigned), st);
    unsigned bits = 0;
    if (err == hipSuccess) {
        const uint64_t want = (n + kThreads - 1) / kThreads;
        hipLaunchKernelGGL(k_absmax, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(kThreads), 0, st, d_values, d_mask, n, d_bits);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpyAsync(&bits, d_bits, sizeof bits, hipMemcpyDeviceToHost, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (!d_scratch_word) (void)hipFree(d_bits);
    PCR_HIP_TRY(err);
    float f;
    static_assert(sizeof f == sizeof bits, "float bits");
    __builtin_memcpy(&f, &bits, sizeof f);
    *h_result = f;
    return PCR_HIP_OK;
}

int pcr_hip_route_count(const pcr_hip_grid* g, const int32_t* row_splits, int nparts,
                        const double* d_x, const double* d_y, const uint8_t* d_mask, uint64_t n,
                        uint8_t* d_dest, unsigned long long* d_counts, pcr_hip_stream s) {
    int rc = validate_grid(g);
    if (rc) return rc;
    PCR_REQUIRE(row_splits && nparts >= 1 && nparts <= kMaxParts, "route_count: 1..64 parts");
    PCR_REQUIRE(d_counts, "route_count: null counts");
    Splits sp;
    sp.n = nparts;
    for (int p = 0; p <= nparts; ++p) {
        PCR_REQUIRE(row_splits[p] >= 0 && row_splits[p] <= g->height && (p == 0 || row_splits[p] >= row_splits[p - 1]),
                    "route_count: row splits must be ascending and inside the grid");
        sp.row[p] = row_splits[p];
    }
    hipStream_t st = static_cast<hipStream_t>(s);
    PCR_HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)nparts * sizeof(unsigned long long), st));
    if (n == 0) return PCR_HIP_OK;
    PCR_REQUIRE(d_x && d_y && d_dest, "route_count: null point array");
    GridDev gd = make_grid_dev(*g);
    gd.mask = d_mask;
    const uint64_t want = (n + kThreads - 1) / kThreads;
    const int blocks = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(k_route_count, dim3(blocks), dim3(kThreads), 0, st, gd, sp, d_x, d_y, n, d_dest, d_counts);
    PCR_HIP_TRY(hipGetLastError());
    return PCR_HIP_OK;
}

int pcr_hip_route_scatter(const uint8_t* d_dest, uint64_t n, int nparts, unsigned long long* d_cursors,
                          int narrays, const void* const* d_src, void* const* d_dst, const int32_t* elem_bytes,
                          pcr_hip_stream s) {
    PCR_REQUIRE(nparts >= 1 && nparts <= kMaxParts, "route_scatter: 1..64 parts");
    PCR_REQUIRE(narrays >= 1 && narrays <= PCR_HIP_MAX_ROUTE_ARRAYS, "route_scatter: 1..8 arrays");
    PCR_REQUIRE(d_cursors && d_src && d_dst && elem_bytes, "route_scatter: null argument");
    if (n == 0) return PCR_HIP_OK;
    PCR_REQUIRE(d_dest, "route_scatter: null owner array");
    PCR_REQUIRE(n < ((uint64_t)1 << 40), "route_scatter: too many points in one call");
    Arrays a;
    a.n = narrays;
    for (int q = 0; q < narrays; ++q) {
        PCR_REQUIRE(d_src[q] && d_dst[q], "route_scatter: null array");
        PCR_REQUIRE(elem_bytes[q] == 4 || elem_bytes[q] == 8, "route_scatter: elements must be 4 or 8 bytes");
        a.src[q] = d_src[q];
        a.dst[q] = d_dst[q];
        a.elem[q] = elem_bytes[q];
    }
    const uint64_t blocks = (n + (uint64_t)kThreads * kPer - 1) / ((uint64_t)kThreads * kPer);
    hipLaunchKernelGGL(k_route_scatter, dim3((unsigned)blocks), dim3(kThreads), 0, static_cast<hipStream_t>(s),
                       d_dest, n, nparts, d_cursors, a);
    PCR_HIP_TRY(hipGetLastError());
    return PCR_HIP_OK;
}

}  // extern "C"
