// runtime.hip -- device / stream / memory / arena half of the C-ABI (include/pcr_hip.h).
#include "common.hpp"

#include <algorithm>
#include <cstring>
#include <new>

namespace pcrhip {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int validate_grid(const pcr_hip_grid* g) {
    PCR_REQUIRE(g != nullptr, "grid: null descriptor");
    PCR_REQUIRE(g->width > 0 && g->height > 0, "grid: dimensions must be positive");
    PCR_REQUIRE(g->tile_width > 0 && g->tile_height > 0, "grid: tile dimensions must be positive");
    PCR_REQUIRE(g->cell_size_x != 0.0 && g->cell_size_y != 0.0, "grid: cell size cannot be zero");
    PCR_REQUIRE((int64_t)g->width * g->height < ((int64_t)1 << 32),
                "grid: width*height must fit 32 bits (reference limit, tile_router.h:24)");
    PCR_REQUIRE(g->own_row0 >= 0 && g->own_row1 <= g->height && g->own_row0 <= g->own_row1,
                "grid: owned row range outside the grid");
    PCR_REQUIRE(g->state_row0 >= 0 && g->state_rows >= 0 && g->state_row0 + g->state_rows <= g->height,
                "grid: state row window outside the grid");
    PCR_REQUIRE(g->state_row0 <= g->own_row0 && g->own_row1 <= g->state_row0 + g->state_rows,
                "grid: state row window must contain the owned rows");
    return PCR_HIP_OK;
}

}  // namespace pcrhip

using namespace pcrhip;

struct pcr_hip_arena {
    char* base = nullptr;
    size_t capacity = 0;
    size_t used = 0;
    size_t high_water = 0;
};

extern "C" {

const char* pcr_hip_last_error(void) { return g_last_error.c_str(); }
int pcr_hip_abi_version(void) { return PCR_HIP_ABI_VERSION; }

// ---- devices ------------------------------------------------------------------
int pcr_hip_device_count(int* count) {
    PCR_REQUIRE(count, "device_count: null out pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }   // "no device" is an answer, not an error
    *count = n;
    return PCR_HIP_OK;
}

int pcr_hip_set_device(int device_id) {
    PCR_HIP_TRY(hipSetDevice(device_id));
    return PCR_HIP_OK;
}

int pcr_hip_get_device(int* device_id) {
    PCR_REQUIRE(device_id, "get_device: null out pointer");
    PCR_HIP_TRY(hipGetDevice(device_id));
    return PCR_HIP_OK;
}

int pcr_hip_device_name(int device_id, char* buf, size_t buf_len) {
    PCR_REQUIRE(buf && buf_len > 0, "device_name: empty buffer");
    hipDeviceProp_t prop;
    PCR_HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    std::snprintf(buf, buf_len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return PCR_HIP_OK;
}

int pcr_hip_mem_info(size_t* free_bytes, size_t* total_bytes) {
    PCR_REQUIRE(free_bytes && total_bytes, "mem_info: null out pointer");
    PCR_HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return PCR_HIP_OK;
}

int pcr_hip_device_synchronize(void) {
    PCR_HIP_TRY(hipDeviceSynchronize());
    return PCR_HIP_OK;
}

// ---- streams / events -----------------------------------------------------------
int pcr_hip_stream_create(pcr_hip_stream* out) {
    PCR_REQUIRE(out, "stream_create: null out pointer");
    hipStream_t s;
    PCR_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = s;
    return PCR_HIP_OK;
}

int pcr_hip_stream_destroy(pcr_hip_stream s) {
    if (s) PCR_HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(s)));
    return PCR_HIP_OK;
}

int pcr_hip_stream_synchronize(pcr_hip_stream s) {
    PCR_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(s)));
    return PCR_HIP_OK;
}

int pcr_hip_event_create(void** out) {
    PCR_REQUIRE(out, "event_create: null out pointer");
    hipEvent_t e;
    PCR_HIP_TRY(hipEventCreate(&e));
    *out = e;
    return PCR_HIP_OK;
}

int pcr_hip_event_destroy(void* ev) {
    if (ev) PCR_HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(ev)));
    return PCR_HIP_OK;
}

int pcr_hip_event_record(void* ev, pcr_hip_stream s) {
    PCR_REQUIRE(ev, "event_record: null event");
    PCR_HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(s)));
    return PCR_HIP_OK;
}

int pcr_hip_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
    PCR_REQUIRE(ev_start && ev_stop && ms, "event_elapsed_ms: null argument");
    PCR_HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(ev_stop)));
    PCR_HIP_TRY(hipEventElapsedTime(ms, static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)));
    return PCR_HIP_OK;
}

// ---- memory ---------------------------------------------------------------------
int pcr_hip_malloc(void** d_ptr, size_t bytes) {
    PCR_REQUIRE(d_ptr, "malloc: null out pointer");
    *d_ptr = nullptr;
    if (bytes == 0) return PCR_HIP_OK;
    PCR_HIP_TRY(hipMalloc(d_ptr, bytes));
    return PCR_HIP_OK;
}

int pcr_hip_free(void* d_ptr) {
    if (d_ptr) PCR_HIP_TRY(hipFree(d_ptr));
    return PCR_HIP_OK;
}

int pcr_hip_host_alloc(void** h_ptr, size_t bytes) {
    PCR_REQUIRE(h_ptr, "host_alloc: null out pointer");
    *h_ptr = nullptr;
    if (bytes == 0) return PCR_HIP_OK;
    PCR_HIP_TRY(hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
    return PCR_HIP_OK;
}

int pcr_hip_host_free(void* h_ptr) {
    if (h_ptr) PCR_HIP_TRY(hipHostFree(h_ptr));
    return PCR_HIP_OK;
}

static int copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, pcr_hip_stream s) {
    if (bytes == 0) return PCR_HIP_OK;
    PCR_REQUIRE(dst && src, "memcpy: null pointer");
    PCR_HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, static_cast<hipStream_t>(s)));
    return PCR_HIP_OK;
}

int pcr_hip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, pcr_hip_stream s) {
    return copy(d_dst, h_src, bytes, hipMemcpyHostToDevice, s);
}
int pcr_hip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, pcr_hip_stream s) {
    return copy(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s);
}
int pcr_hip_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, pcr_hip_stream s) {
    return copy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, s);
}

}  // extern "C"

// ---- the streaming-copy yardstick ------------------------------------------------------------------------------
// A hand-written float4 copy (16 bytes per lane and access, four independent accesses in flight per lane): the shape
// /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s for.  bench.py reports it as `measured_copy_GBps`, the practical
// roof of THIS box, next to torch's copy_ (the runtime's blit kernel) and the 8 TB/s data-sheet peak.
namespace {
typedef float pcr_f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256)
k_copy_f4(pcr_f4v* __restrict__ dst, const pcr_f4v* __restrict__ src, size_t n4) {
    // a workgroup owns 4 x 256 consecutive float4 (16 KB); lane t reads t, t + 256, t + 512, t + 768: every wave
    // instruction is one contiguous 1-KB request
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    pcr_f4v v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n4) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n4) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
    }
}
}  // namespace

extern "C" {

int pcr_hip_copy_kernel(void* d_dst, const void* d_src, size_t bytes, int nontemporal, pcr_hip_stream s) {
    if (bytes == 0) return PCR_HIP_OK;
    PCR_REQUIRE(d_dst && d_src, "copy_kernel: null pointer");
    PCR_REQUIRE(((reinterpret_cast<uintptr_t>(d_dst) | reinterpret_cast<uintptr_t>(d_src) | bytes) & 15) == 0,
                "copy_kernel: pointers and size must be multiples of 16 bytes");
    const size_t n4 = bytes / 16;
    const size_t blocks = (n4 + 1023) / 1024;
    PCR_REQUIRE(blocks <= 0x7FFFFFFFull, "copy_kernel: more than 2^31 workgroups");
    hipStream_t st = static_cast<hipStream_t>(s);
    if (nontemporal)
        hipLaunchKernelGGL(k_copy_f4<true>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<pcr_f4v*>(d_dst), static_cast<const pcr_f4v*>(d_src), n4);
    else
        hipLaunchKernelGGL(k_copy_f4<false>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<pcr_f4v*>(d_dst), static_cast<const pcr_f4v*>(d_src), n4);
    PCR_HIP_TRY(hipGetLastError());
    return PCR_HIP_OK;
}

int pcr_hip_memset(void* d_ptr, int byte_value, size_t bytes, pcr_hip_stream s) {
    if (bytes == 0) return PCR_HIP_OK;
    PCR_REQUIRE(d_ptr, "memset: null pointer");
    PCR_HIP_TRY(hipMemsetAsync(d_ptr, byte_value, bytes, static_cast<hipStream_t>(s)));
    return PCR_HIP_OK;
}

// ---- arena (bump allocator, 256-B aligned; the MemoryPool contract) ---------------
int pcr_hip_arena_create(pcr_hip_arena** out, size_t bytes) {
    PCR_REQUIRE(out, "arena_create: null out pointer");
    PCR_REQUIRE(bytes > 0, "arena_create: size must be positive");
    auto* a = new (std::nothrow) pcr_hip_arena();
    if (!a) return fail(PCR_HIP_OUT_OF_MEMORY, "arena_create: host allocation failed");
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&a->base), bytes);
    if (e != hipSuccess) {
        delete a;
        return fail(e == hipErrorOutOfMemory ? PCR_HIP_OUT_OF_MEMORY : PCR_HIP_CUDA_ERROR,
                    std::string("arena_create: ") + hipGetErrorString(e));
    }
    a->capacity = bytes;
    *out = a;
    return PCR_HIP_OK;
}

int pcr_hip_arena_destroy(pcr_hip_arena* a) {
    if (!a) return PCR_HIP_OK;
    hipError_t e = a->base ? hipFree(a->base) : hipSuccess;
    delete a;
    if (e != hipSuccess) return fail(PCR_HIP_CUDA_ERROR, std::string("arena_destroy: ") + hipGetErrorString(e));
    return PCR_HIP_OK;
}

int pcr_hip_arena_alloc(pcr_hip_arena* a, size_t bytes, void** d_ptr) {
    PCR_REQUIRE(a && d_ptr, "arena_alloc: null argument");
    size_t start = (a->used + 255) & ~size_t(255);
    if (start + bytes > a->capacity) {
        *d_ptr = nullptr;
        return fail(PCR_HIP_OUT_OF_MEMORY, "arena_alloc: pool exhausted");
    }
    *d_ptr = a->base + start;
    a->used = start + bytes;
    a->high_water = std::max(a->high_water, a->used);
    return PCR_HIP_OK;
}

int pcr_hip_arena_reset(pcr_hip_arena* a) {
    PCR_REQUIRE(a, "arena_reset: null arena");
    a->used = 0;
    return PCR_HIP_OK;
}

int pcr_hip_arena_stats(const pcr_hip_arena* a, size_t* capacity, size_t* used, size_t* high_water) {
    PCR_REQUIRE(a, "arena_stats: null arena");
    if (capacity) *capacity = a->capacity;
    if (used) *used = a->used;
    if (high_water) *high_water = a->high_water;
    return PCR_HIP_OK;
}

}  // extern "C"
