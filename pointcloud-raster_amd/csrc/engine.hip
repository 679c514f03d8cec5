// engine.hip -- pcr_hip_engine_* and pcr_hip_scatter_*: argument checking, path choice
// (direct global atomics vs binned LDS tiles), bookkeeping.  Kernels live in
// scatter_direct.hip / scatter_binned.hip.
#include "engine.hpp"

#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

using namespace pcrhip;

namespace {

// One grow-only scratch arena per device, shared by every engine on it (the reference's
// MemoryPool role, src/engine/memory_pool.cu:24-59; the block IS a pcr_hip_arena, the same bump
// allocator the C-ABI exports).  A scatter borrows the whole arena for the kernels it enqueues:
//   * ownership is exclusive from ensure_scratch to release_scratch (borrow_mu is held across the two
//     calls), so two host threads driving two pipelines on one device can never carve the same bytes;
//   * ordering between borrowers on different streams is by event: the borrower's stream waits for the
//     previous borrower's last kernel, so nothing synchronizes the host in steady state and a fresh
//     pipeline never pays a multi-GB hipMalloc on its first ingest;
//   * the block is only ever replaced by its current owner, after a device-wide synchronize.
struct SharedScratch {
    std::mutex borrow_mu;                 // held while an engine has the arena borrowed
    std::mutex mu;                        // bookkeeping below
    pcr_hip_arena* arena = nullptr;
    hipEvent_t last = nullptr;
    bool has_last = false;
    uint64_t borrows = 0, grows = 0;
    // moment path: the tap tables of the last glyph spec stay on the device (touched only while borrow_mu is held).
    // A fresh pipeline's first Gaussian scatter then uploads nothing: the bench builds one pipeline per step, and a
    // pageable host-to-device copy inside every step made the step time depend on what else the host was doing.
    struct Taps {
        int K = -1, r = -1;
        float sx = 0.f, sy = 0.f;
        float* d = nullptr;
        size_t cap = 0;                   // floats
        std::vector<float> host;
    } taps;
};
constexpr int kMaxDevices = 64;
SharedScratch g_scratch[kMaxDevices];

}  // namespace

namespace pcrhip {

int ensure_scratch(pcr_hip_engine* e, size_t bytes) {
    PCR_REQUIRE(e->device >= 0 && e->device < kMaxDevices, "scratch: device ordinal out of range");
    SharedScratch& sp = g_scratch[e->device];
    const bool fresh = !e->scratch_borrowed;
    if (fresh) sp.borrow_mu.lock();                       // released by release_scratch (same host thread)
    std::lock_guard<std::mutex> lock(sp.mu);
    auto bail = [&](int rc) { if (fresh) sp.borrow_mu.unlock(); return rc; };
    if (!sp.last && hipEventCreateWithFlags(&sp.last, hipEventDisableTiming) != hipSuccess)
        return bail(fail(PCR_HIP_CUDA_ERROR, "scratch: cannot create the hand-over event"));
    size_t cap = 0;
    if (sp.arena) (void)pcr_hip_arena_stats(sp.arena, &cap, nullptr, nullptr);
    if (bytes > cap) {
        // growth only (first ingest of a larger size): every earlier user must be done with the old block,
        // and nobody else can hold it (we own borrow_mu)
        if (hipDeviceSynchronize() != hipSuccess) return bail(fail(PCR_HIP_CUDA_ERROR, "scratch: device synchronize failed"));
        if (sp.arena) (void)pcr_hip_arena_destroy(sp.arena);
        sp.arena = nullptr;
        sp.has_last = false;
        int rc = pcr_hip_arena_create(&sp.arena, bytes + bytes / 8);
        if (rc) return bail(rc);
        ++sp.grows;
    }
    // one scatter = one allocation that spans what its passes carve up
    void* base = nullptr;
    (void)pcr_hip_arena_reset(sp.arena);
    int rc = pcr_hip_arena_alloc(sp.arena, bytes, &base);
    if (rc) return bail(rc);
    if (fresh) {
        if (sp.has_last && hipStreamWaitEvent(e->stream, sp.last, 0) != hipSuccess)
            return bail(fail(PCR_HIP_CUDA_ERROR, "scratch: cannot order after the previous borrower"));
        ++sp.borrows;
    }
    (void)pcr_hip_arena_stats(sp.arena, &cap, nullptr, nullptr);
    e->d_scratch = static_cast<char*>(base);
    e->scratch_cap = cap;
    e->scratch_borrowed = true;
    return PCR_HIP_OK;
}

int shared_taps(pcr_hip_engine* e, int K, int r, float sx, float sy,
                void (*fill)(std::vector<float>&, int, int, float, float), const float** d_taps, size_t* count) {
    PCR_REQUIRE(e->scratch_borrowed, "shared_taps: the scratch must be borrowed");
    auto& t = g_scratch[e->device].taps;
    if (!(t.d && t.K == K && t.r == r && t.sx == sx && t.sy == sy)) {
        // every kernel that reads the old tables was enqueued by an earlier borrower: this stream is ordered after
        // it (ensure_scratch); the host copy may still feed an earlier upload of this stream
        PCR_HIP_TRY(hipStreamSynchronize(e->stream));
        std::vector<float> next;
        fill(next, K, r, sx, sy);
        if (next.size() > t.cap) {
            PCR_HIP_TRY(hipDeviceSynchronize());
            if (t.d) (void)hipFree(t.d);
            t.d = nullptr;
            t.cap = 0;
            PCR_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t.d), next.size() * sizeof(float)));
            t.cap = next.size();
        }
        t.host.swap(next);
        t.K = -1;                                                   // invalid until the upload is enqueued
        PCR_HIP_TRY(hipMemcpyAsync(t.d, t.host.data(), t.host.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
        t.K = K; t.r = r; t.sx = sx; t.sy = sy;
    }
    *d_taps = t.d;
    *count = t.host.size();
    return PCR_HIP_OK;
}

// Called after a scatter has enqueued its last kernel.
void release_scratch(pcr_hip_engine* e) {
    if (!e->scratch_borrowed) return;
    SharedScratch& sp = g_scratch[e->device];
    {
        std::lock_guard<std::mutex> lock(sp.mu);
        if (sp.last && hipEventRecord(sp.last, e->stream) == hipSuccess) sp.has_last = true;
        e->scratch_borrowed = false;
        e->d_scratch = nullptr;
    }
    sp.borrow_mu.unlock();
}

}  // namespace pcrhip

namespace {

// Every entry point that launches or allocates runs on the engine's device, whatever device the
// calling thread had current (torch, or a second pipeline on another GPU, may have changed it).
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int check_planes(uint32_t mask, const pcr_hip_planes* p, uint32_t allowed, PlanesDev& out) {
    PCR_REQUIRE(p != nullptr, "scatter: null planes");
    PCR_REQUIRE(mask != 0 && (mask & ~allowed) == 0, "scatter: plane mask not supported for this glyph");
    PCR_REQUIRE(!(mask & PCR_HIP_PLANE_SUM) || p->d_sum, "scatter: SUM plane requested but d_sum is null");
    PCR_REQUIRE(!(mask & PCR_HIP_PLANE_WGT) || p->d_wgt, "scatter: WGT plane requested but d_wgt is null");
    PCR_REQUIRE(!(mask & PCR_HIP_PLANE_MAX) || p->d_max, "scatter: MAX plane requested but d_max is null");
    PCR_REQUIRE(!(mask & PCR_HIP_PLANE_MIN) || p->d_min, "scatter: MIN plane requested but d_min is null");
    out.sum = p->d_sum; out.wgt = p->d_wgt; out.mx = p->d_max; out.mn = p->d_min;
    return PCR_HIP_OK;
}

int begin_scatter(pcr_hip_engine* e, uint64_t n) {
    e->stats.points_in = n;
    e->stats.points_valid = 0;
    e->stats.lds_tile_w = e->stats.lds_tile_h = e->stats.lds_apron = e->stats.num_bins = 0;
    e->stats_scatter_chunk = 0;
    PCR_HIP_TRY(hipMemsetAsync(e->d_counters, 0, 8 * sizeof(unsigned long long), e->stream));
    return PCR_HIP_OK;
}

}  // namespace

namespace pcrhip {

// State initialisation inside the scatter (the reference initialises tile state inside ingest too, pipeline.cpp:688-691):
// only the paths that cannot define every cell themselves pay for it.
int fill_identity(pcr_hip_engine* e, uint32_t mask, const PlanesDev& pl) {
    ScopedKernelTimer t(e, "k_state_init");
    const int64_t cells = (int64_t)e->gd.st_rows * e->gd.W;
    int rc = PCR_HIP_OK;
    if ((mask & PCR_HIP_PLANE_SUM) && (rc = pcr_hip_plane_fill(pl.sum, 0.0f, cells, e->stream)) != PCR_HIP_OK) return rc;
    if ((mask & PCR_HIP_PLANE_WGT) && (rc = pcr_hip_plane_fill(pl.wgt, 0.0f, cells, e->stream)) != PCR_HIP_OK) return rc;
    if ((mask & PCR_HIP_PLANE_MAX) && (rc = pcr_hip_plane_fill(pl.mx, -FLT_MAX, cells, e->stream)) != PCR_HIP_OK) return rc;
    if ((mask & PCR_HIP_PLANE_MIN) && (rc = pcr_hip_plane_fill(pl.mn, FLT_MAX, cells, e->stream)) != PCR_HIP_OK) return rc;
    return rc;
}

}  // namespace pcrhip

extern "C" {

int pcr_hip_engine_create(pcr_hip_engine** out, const pcr_hip_grid* g, size_t scratch_bytes, pcr_hip_stream s) {
    PCR_REQUIRE(out, "engine_create: null out pointer");
    *out = nullptr;
    int rc = validate_grid(g);
    if (rc) return rc;
    auto* e = new (std::nothrow) pcr_hip_engine();
    if (!e) return fail(PCR_HIP_OUT_OF_MEMORY, "engine_create: host allocation failed");
    e->grid = *g;
    e->gd = make_grid_dev(*g);
    e->stream = static_cast<hipStream_t>(s);
    e->max_bins = kMaxBins;
    if (const char* dbg = std::getenv("PCR_HIP_DEBUG_MAX_BINS")) {
        const int v = std::atoi(dbg);
        if (v >= 1 && v < kMaxBins) e->max_bins = v;
    }
    if (const char* dbg = std::getenv("PCR_HIP_DEBUG_TWO_LEVEL")) e->two_level = std::atoi(dbg) != 0;
    hipError_t err = hipGetDevice(&e->device);
    hipDeviceProp_t prop;
    if (err == hipSuccess) err = hipGetDeviceProperties(&prop, e->device);
    if (err == hipSuccess) {
        e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        e->ntiles = e->gd.tiles_x * e->gd.tiles_y;
        err = hipMalloc(reinterpret_cast<void**>(&e->d_touched), (size_t)e->ntiles * sizeof(uint32_t));
    }
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&e->d_counters), 8 * sizeof(unsigned long long));
    if (err == hipSuccess) err = hipMemsetAsync(e->d_touched, 0, (size_t)e->ntiles * sizeof(uint32_t), e->stream);
    if (err == hipSuccess) err = hipMemsetAsync(e->d_counters, 0, 8 * sizeof(unsigned long long), e->stream);
    if (err == hipSuccess && scratch_bytes) {
        // pre-size the device-wide arena at create time (outside any ingest)
        if (ensure_scratch(e, scratch_bytes) != PCR_HIP_OK) err = hipErrorOutOfMemory;
        else release_scratch(e);
    }
    if (err != hipSuccess) {
        std::string msg = std::string("engine_create: ") + hipGetErrorString(err);
        pcr_hip_engine_destroy(e);
        return fail(err == hipErrorOutOfMemory ? PCR_HIP_OUT_OF_MEMORY : PCR_HIP_CUDA_ERROR, msg);
    }
    *out = e;
    return PCR_HIP_OK;
}

int pcr_hip_engine_destroy(pcr_hip_engine* e) {
    if (!e) return PCR_HIP_OK;
    DeviceGuard dev(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (auto& p : e->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (e->d_touched) (void)hipFree(e->d_touched);
    if (e->d_counters) (void)hipFree(e->d_counters);
    delete e;              // the scratch arena is device-wide and outlives engines
    return PCR_HIP_OK;
}

int pcr_hip_device_scratch_stats(int device, size_t* capacity, size_t* high_water, uint64_t* borrows, uint64_t* grows) {
    PCR_REQUIRE(device >= 0 && device < kMaxDevices, "device_scratch_stats: device ordinal out of range");
    SharedScratch& sp = g_scratch[device];
    std::lock_guard<std::mutex> lock(sp.mu);
    size_t cap = 0, hw = 0;
    if (sp.arena) (void)pcr_hip_arena_stats(sp.arena, &cap, nullptr, &hw);
    if (capacity) *capacity = cap;
    if (high_water) *high_water = hw;
    if (borrows) *borrows = sp.borrows;
    if (grows) *grows = sp.grows;
    return PCR_HIP_OK;
}

int pcr_hip_engine_set_path(pcr_hip_engine* e, int path) {
    PCR_REQUIRE(e, "engine_set_path: null engine");
    PCR_REQUIRE(path >= 0 && path <= 3,
                "engine_set_path: path must be 0 (auto), 1 (direct), 2 (binned) or 3 (moments, Gaussian only)");
    e->forced_path = path;
    return PCR_HIP_OK;
}

int pcr_hip_engine_planes_fresh(pcr_hip_engine* e, int fresh) {
    PCR_REQUIRE(e, "engine_planes_fresh: null engine");
    PCR_REQUIRE(fresh >= 0 && fresh <= 2, "engine_planes_fresh: 0 (accumulate), 1 (identity-filled) or 2 (undefined) expected");
    e->planes_fresh = fresh;
    return PCR_HIP_OK;
}

int pcr_hip_engine_finalize_with_scatter(pcr_hip_engine* e, int n_out, const int* rtypes, float* const* d_outs,
                                         uint32_t* d_bands_done) {
    PCR_REQUIRE(e, "engine_finalize_with_scatter: null engine");
    e->fused_outs.n = 0;
    e->fused_done = nullptr;
    if (n_out == 0) return PCR_HIP_OK;                    // withdraws the hint
    PCR_REQUIRE(rtypes && d_outs && d_bands_done, "engine_finalize_with_scatter: null argument");
    PCR_REQUIRE(n_out >= 1 && n_out <= PCR_HIP_MAX_FINALIZE_OUTPUTS, "engine_finalize_with_scatter: 1..8 outputs");
    for (int i = 0; i < n_out; ++i) {
        PCR_REQUIRE(rtypes[i] >= PCR_HIP_SUM && rtypes[i] <= PCR_HIP_COUNT, "pipeline: unknown reduction type");
        PCR_REQUIRE(d_outs[i] && (reinterpret_cast<uintptr_t>(d_outs[i]) & 15) == 0,
                    "engine_finalize_with_scatter: bands must be non-null and 16-byte aligned");
        e->fused_outs.rtype[i] = rtypes[i];
        e->fused_outs.out[i] = d_outs[i];
    }
    e->fused_outs.n = n_out;
    e->fused_done = d_bands_done;
    return PCR_HIP_OK;
}

int pcr_hip_engine_finalize_taken(const pcr_hip_engine* e) { return e && e->fused_taken ? 1 : 0; }

int pcr_hip_engine_stats(const pcr_hip_engine* e, pcr_hip_scatter_stats* out) {
    PCR_REQUIRE(e && out, "engine_stats: null argument");
    unsigned long long c[8] = {0};
    DeviceGuard dev(e->device);
    PCR_HIP_TRY(hipMemcpyAsync(c, e->d_counters, sizeof c, hipMemcpyDeviceToHost, e->stream));
    PCR_HIP_TRY(hipStreamSynchronize(e->stream));
    *out = e->stats;
    out->points_valid = c[0];
    out->scatter_chunk = e->stats.path == 1 ? e->stats_scatter_chunk : 0;
    return PCR_HIP_OK;
}

int pcr_hip_engine_tile_touched(pcr_hip_engine* e, uint32_t** d_tile_touched, int32_t* tiles_x, int32_t* tiles_y) {
    PCR_REQUIRE(e && d_tile_touched, "engine_tile_touched: null argument");
    *d_tile_touched = e->d_touched;
    if (tiles_x) *tiles_x = e->gd.tiles_x;
    if (tiles_y) *tiles_y = e->gd.tiles_y;
    return PCR_HIP_OK;
}

int pcr_hip_engine_set_point_mask(pcr_hip_engine* e, const uint8_t* d_mask) {
    PCR_REQUIRE(e, "engine_set_point_mask: null engine");
    e->gd.mask = d_mask;
    return PCR_HIP_OK;
}

int pcr_hip_engine_profile_enable(pcr_hip_engine* e, int on) {
    PCR_REQUIRE(e, "engine_profile_enable: null engine");
    e->profiling = on != 0;
    return PCR_HIP_OK;
}

int pcr_hip_engine_profile_only(pcr_hip_engine* e, const char* kernel_name) {
    PCR_REQUIRE(e, "engine_profile_only: null engine");
    e->profile_only = kernel_name ? kernel_name : "";
    return PCR_HIP_OK;
}

int pcr_hip_engine_profile_read(pcr_hip_engine* e, pcr_hip_kernel_time* out, int capacity, int* count, int reset) {
    PCR_REQUIRE(e && count, "engine_profile_read: null argument");
    DeviceGuard dev(e->device);
    for (auto& p : e->pending) {
        float ms = 0.0f;
        hipError_t err = hipEventSynchronize(p.b);
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, p.a, p.b);
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
        if (err == hipSuccess) {
            auto& slot = e->kernel_ms[p.name];
            slot.first += 1;
            slot.second += ms;
        }
    }
    e->pending.clear();
    int n = 0;
    for (const auto& kv : e->kernel_ms) {
        if (out && n < capacity) {
            std::snprintf(out[n].name, sizeof out[n].name, "%s", kv.first.c_str());
            out[n].launches = kv.second.first;
            out[n].total_ms = kv.second.second;
        }
        ++n;
    }
    *count = n;
    if (reset) e->kernel_ms.clear();
    return PCR_HIP_OK;
}

int pcr_hip_scatter_point(pcr_hip_engine* e, uint32_t plane_mask, const pcr_hip_planes* planes,
                          const double* d_x, const double* d_y, const float* d_value, uint64_t n) {
    PCR_REQUIRE(e, "scatter_point: null engine");
    struct HintGuard {                                    // the fused-finalize hint covers this call, however it ends
        pcr_hip_engine* e;
        ~HintGuard() { e->fused_outs.n = 0; e->fused_done = nullptr; }
    } hint_guard{e};
    e->fused_taken = false;
    PlanesDev pl;
    int rc = check_planes(plane_mask, planes, 15u, pl);
    if (rc) return rc;
    if (n == 0) return PCR_HIP_OK;                       // empty cloud is a no-op (pipeline.cpp:284-287)
    PCR_REQUIRE(n < ((uint64_t)1 << 40), "scatter_point: too many points in one call");
    PCR_REQUIRE(d_x && d_y, "scatter_point: null coordinate array");
    PCR_REQUIRE(d_value || plane_mask == PCR_HIP_PLANE_WGT, "scatter_point: null value array");
    DeviceGuard dev(e->device);
    rc = begin_scatter(e, n);
    if (rc) return rc;
    bool can_bin = binned_point_supported(e, plane_mask);
    if (e->forced_path == 2 && !can_bin)
        return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_point: binned path forced but not applicable to this grid");
    if (e->forced_path == 1 || !can_bin) {
        if (e->planes_fresh == 2 && (rc = fill_identity(e, plane_mask, pl)) != PCR_HIP_OK) return rc;
        e->planes_fresh = 0;
        return direct_point(e, plane_mask, pl, d_x, d_y, d_value, n);
    }
    rc = binned_point(e, plane_mask, pl, d_x, d_y, d_value, n);
    e->planes_fresh = 0;                                  // the hint covers one scatter
    release_scratch(e);
    return rc;
}

int pcr_hip_scatter_glyph(pcr_hip_engine* e, const pcr_hip_glyph* glyph, uint32_t plane_mask,
                          const pcr_hip_planes* planes,
                          const double* d_x, const double* d_y, const float* d_value, uint64_t n) {
    PCR_REQUIRE(e && glyph, "scatter_glyph: null argument");
    if (glyph->type == PCR_HIP_GLYPH_POINT)
        return fail(PCR_HIP_INVALID_ARGUMENT, "accumulate_glyph: Point glyph should use regular accumulate()");
    if (glyph->type != PCR_HIP_GLYPH_LINE && glyph->type != PCR_HIP_GLYPH_GAUSSIAN)
        return fail(PCR_HIP_NOT_IMPLEMENTED, "glyph: unknown glyph type");
    if (plane_mask & (PCR_HIP_PLANE_MAX | PCR_HIP_PLANE_MIN))
        return fail(PCR_HIP_NOT_IMPLEMENTED,
                    "glyph splatting only supports WeightedAverage, Average, Sum, or Count reduction types");
    PlanesDev pl;
    int rc = check_planes(plane_mask, planes, 3u, pl);
    if (rc) return rc;
    if (n == 0) return PCR_HIP_OK;
    PCR_REQUIRE(n < ((uint64_t)1 << 40), "scatter_glyph: too many points in one call");
    PCR_REQUIRE(d_x && d_y && d_value, "scatter_glyph: null point array");
    GlyphDev gl;
    gl.type = glyph->type;
    gl.def_direction = glyph->default_direction;
    gl.def_half_length = glyph->default_half_length;
    gl.def_sigma_x = glyph->default_sigma_x;
    gl.def_sigma_y = glyph->default_sigma_y;
    gl.def_rotation = glyph->default_rotation;
    gl.max_radius = glyph->max_radius_cells;
    gl.direction = glyph->d_direction;
    gl.half_length = glyph->d_half_length;
    gl.sigma_x = glyph->d_sigma_x;
    gl.sigma_y = glyph->d_sigma_y;
    gl.rotation = glyph->d_rotation;
    DeviceGuard dev(e->device);
    const bool undefined = e->planes_fresh == 2;
    e->planes_fresh = 0;                                  // the hint covers one scatter
    rc = begin_scatter(e, n);
    if (rc) return rc;
    if (e->forced_path == 3 || e->forced_path == 0) {
        bool can_mom = moments_supported(e, gl, plane_mask);
        if (e->forced_path == 3 && !can_mom && gl.type == PCR_HIP_GLYPH_GAUSSIAN)
            return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_glyph: moment path forced but not applicable to this glyph");
        if (can_mom) {
            // (undefined planes: the moment path's row pass stores every cell when one window covers the state window)
            rc = moments_gauss(e, gl, plane_mask, pl, d_x, d_y, d_value, n, undefined);
            release_scratch(e);
            return rc;
        }
    }
    // the tile merges accumulate (atomics, overlapping aprons): undefined planes are given their identity values first
    if (undefined && (rc = fill_identity(e, plane_mask, pl)) != PCR_HIP_OK) return rc;
    bool can_bin = binned_glyph_supported(e, gl, plane_mask);
    if (e->forced_path == 2 && !can_bin)
        return fail(PCR_HIP_INVALID_ARGUMENT, "scatter_glyph: binned path forced but not applicable");
    if (e->forced_path == 1 || !can_bin) return direct_glyph(e, gl, plane_mask, pl, d_x, d_y, d_value, n);
    if (cells_gauss_supported(e, gl, plane_mask)) rc = cells_gauss(e, gl, plane_mask, pl, d_x, d_y, d_value, n);
    else rc = binned_glyph(e, gl, plane_mask, pl, d_x, d_y, d_value, n);
    release_scratch(e);
    return rc;
}

}  // extern "C"
