// pipeline.cpp -- Pipeline::{create, ingest, finalize, ...}: host orchestration of the HIP engine.
//
// Same observable contract as the reference's src/engine/pipeline.cpp (validation and error
// strings :365-378, 500-508; empty cloud no-op :284-287; per-ingest progress callback and
// "cancelled by user" :753-767; band naming :1175-1186; NaN for untouched tiles :1204-1222;
// state survives finalize), but a different machine underneath:
//   * no router / sort / per-tile batches / tile manager: reductions that read the same value
//     channel through the same glyph share ONE pass over the points and ONE set of device
//     planes (sum, weight, max, min) in grid layout, resident in HBM for the pipeline's life;
//   * finalize runs on the device; only finalized bands cross PCIe (or stay in HBM);
//   * every device action goes through the C-ABI of include/pcr_hip.h.
#include "pcr/engine/pipeline.h"

#include "buffer.h"
#include "host_pipeline.h"
#include "pipeline_common.h"
#include "pcr/core/grid.h"
#include "pcr/core/point_cloud.h"
#include "pcr/io/grid_io.h"
#include "pcr/io/point_cloud_io.h"
#include "pcr/io/tile_state_io.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <limits>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include <unistd.h>

namespace pcr {

namespace {

thread_local std::string g_create_error;

using detail::registered;
using detail::glyph_reduction_ok;
using detail::planes_for;
using detail::same_glyph;
using detail::kPlaneBits;

}  // namespace

const std::string& pipeline_create_error() { return g_create_error; }

struct Pipeline::Impl {
    struct Group {                       // one pass over the points
        std::string value_channel;
        GlyphSpec glyph;
        uint32_t mask = 0;
        detail::Buffer planes[4];
        pcr_hip_planes view{};
        bool fresh = true;               // nothing has been accumulated yet: the first Point merge may store
        bool defined = false;            // the planes hold values (identity or accumulated).  They are NOT filled at create:
                                         // the first scatter defines them inside ingest, as the reference initialises its tile
                                         // state inside ingest (pipeline.cpp:688-691) -- see define_planes()
        bool bands_with_scatter = false; // the scatter that defined the planes was asked to store this group's finished bands
                                         // too (pcr_hip_engine_finalize_with_scatter) and nothing has touched the planes or the
                                         // touched flags since: finalize() skips the group's kernel when the device agrees
    };
    struct Output {                      // one ReductionSpec -> one band
        int group = 0;
        ReductionType type = ReductionType::Sum;
        std::string band_name;
    };

    PipelineConfig cfg;
    ProgressCallback callback;
    pcr_hip_grid hg{};
    pcr_hip_engine* engine = nullptr;
    pcr_hip_stream stream = nullptr;
    bool own_stream = false;
    std::vector<Group> groups;
    std::vector<Output> outputs;
    std::vector<detail::Buffer> d_bands;     // finalized bands on the device (result_location == Host)
    detail::Buffer d_bands_done;             // one device word per group: set by a scatter that stored the group's bands
    bool state_shared = false;               // plane / touched-flag pointers have left the pipeline: never finalize with a scatter
    std::unique_ptr<Grid> result;
    bool finalized = false;                  // result() is null until the first finalize, as in the reference
    std::map<std::string, detail::Buffer> staging;   // device copies of host-resident arrays, grow-only
    int halo = 0;
    size_t collections = 0;
    size_t points = 0;
    bool continue_on_host = false;           // init() failed where the reference carries on in CPU mode (see init)
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();

    ~Impl() {
        if (engine) pcr_hip_engine_destroy(engine);
        groups.clear();
        d_bands.clear();
        staging.clear();
        result.reset();
        if (own_stream && stream) pcr_hip_stream_destroy(stream);
    }

    int own_rows() const { return hg.own_row1 - hg.own_row0; }

    // Bands a scatter may have stored are stale as soon as anything else writes the planes or the touched flags.
    void bands_stale() { for (auto& gr : groups) gr.bands_with_scatter = false; }
    float* band_device(size_t r) {
        return cfg.result_location == MemoryLocation::Device ? result->band_f32((int)r) : static_cast<float*>(d_bands[r].data());
    }
    // The first Point scatter of a group defines its planes; when this device owns the whole state window (no halo) and
    // nobody outside holds pointers into the state, that scatter's tile pass is asked to store the group's finished bands too
    // -- finalize after a pipeline's only ingest then has nothing to read back (reference: finalize_result re-reads every
    // tile's state, src/engine/pipeline.cpp:1154-1286).
    bool offer_bands(size_t gi) {
        const Group& gr = groups[gi];
        if (!cfg.finalize_with_first_ingest || gr.defined || gr.glyph.type != GlyphType::Point || state_shared || !result ||
            !d_bands_done.data()) return false;
        if (hg.own_row0 != hg.state_row0 || own_rows() != hg.state_rows || own_rows() <= 0) return false;
        int types[PCR_HIP_MAX_FINALIZE_OUTPUTS];
        float* dsts[PCR_HIP_MAX_FINALIZE_OUTPUTS];
        int n = 0;
        for (size_t r = 0; r < outputs.size(); ++r) {
            if (outputs[r].group != (int)gi) continue;
            if (n == PCR_HIP_MAX_FINALIZE_OUTPUTS) return false;
            types[n] = static_cast<int>(outputs[r].type);
            dsts[n++] = band_device(r);
        }
        if (n == 0) return false;
        return pcr_hip_engine_finalize_with_scatter(engine, n, types, dsts, static_cast<uint32_t*>(d_bands_done.data()) + gi) == PCR_HIP_OK;
    }

    Status init() {
        const GridConfig& g = cfg.grid;
        if (g.width <= 0 || g.height <= 0)
            return Status::error(StatusCode::InvalidArgument, "pipeline: grid dimensions must be positive");
        if (g.tile_width <= 0 || g.tile_height <= 0)
            return Status::error(StatusCode::InvalidArgument, "pipeline: tile dimensions must be positive");
        for (const auto& r : cfg.reductions)
            if (!registered(r.type))
                return Status::error(StatusCode::InvalidArgument, "pipeline: unknown reduction type");

        // The reference's GPU-initialisation matrix (src/engine/pipeline.cpp:108-162), with its exact messages and status
        // codes.  Where the reference goes on in CPU mode it prints a Warning / Info line on stderr first; so does this build,
        // and then Pipeline::create continues on the host engine (host_pipeline.h) -- never silently: the line is printed, and
        // Pipeline::engine() says which engine a pipeline runs on.  PCR_REQUIRE_GPU_ENGINE=1 in the environment turns every
        // such fallback into an error (GPU test suites and benchmarks set it: a CPU result must not pass for a GPU one).
        auto no_cpu_engine = [this](const char* how) {
            continue_on_host = true;
            return Status::error(StatusCode::NotImplemented,
                std::string("pipeline: ") + how + ", and PCR_REQUIRE_GPU_ENGINE forbids it");
        };
        int ndev = 0;
        pcr_hip_device_count(&ndev);
        if (ndev <= 0) {
            const std::string msg = "No CUDA-capable GPU detected";
            if (cfg.exec_mode != ExecutionMode::GPU) {                                                  // Auto, Hybrid
                std::fprintf(stderr, "Info: %s - using CPU mode\n", msg.c_str());                       // pipeline.cpp:130
                return no_cpu_engine("no GPU was found and the reference would use its CPU mode");
            }
            if (cfg.gpu_require_strict)
                return Status::error(StatusCode::CudaError, msg + " - GPU mode requested but no GPU available");   // :118-119
            if (cfg.gpu_fallback_to_cpu) {
                std::fprintf(stderr, "Warning: %s - falling back to CPU mode\n", msg.c_str());          // :122
                return no_cpu_engine("the reference would fall back to its CPU mode here");
            }
            return Status::error(StatusCode::CudaError, msg + " - GPU required but not available");         // :125-126
        }
        Status s = Status::success();
        if (cfg.cuda_device_id >= 0) {
            const int rc = cfg.cuda_device_id < ndev ? pcr_hip_set_device(cfg.cuda_device_id) : PCR_HIP_INVALID_ARGUMENT;
            if (rc != PCR_HIP_OK) {
                const std::string err_msg = "Failed to set CUDA device " + std::to_string(cfg.cuda_device_id) + ": " +
                                            (cfg.cuda_device_id < ndev ? std::string(pcr_hip_last_error()) : std::string("invalid device ordinal"));   // :136-137
                if (cfg.gpu_require_strict) return Status::error(StatusCode::CudaError, err_msg);          // :141
                if (cfg.gpu_fallback_to_cpu) {
                    std::fprintf(stderr, "Warning: %s - falling back to CPU mode\n", err_msg.c_str());   // :143
                    return no_cpu_engine("the reference would fall back to its CPU mode here");
                }
                return Status::error(StatusCode::CudaError, err_msg);                                      // :146
            }
        } else {
            return Status::error(StatusCode::InvalidArgument, "pipeline: invalid device id");
        }
        {
            // pipeline.cpp:149-160, once per device and process (the reference prints it at every create; a bench that
            // creates a pipeline per step would bury its own output)
            static std::mutex mu;
            static std::set<int> announced;
            std::lock_guard<std::mutex> lock(mu);
            if (announced.insert(cfg.cuda_device_id).second) {
                const std::string name = cuda_device_name(cfg.cuda_device_id);
                size_t free_mem = 0, total_mem = 0;
                if (cuda_get_memory_info(&free_mem, &total_mem, cfg.cuda_device_id))
                    std::fprintf(stderr, "Info: Using GPU %d: %s (%.1f GB free / %.1f GB total)\n", cfg.cuda_device_id, name.c_str(),
                                 free_mem / (1024.0 * 1024.0 * 1024.0), total_mem / (1024.0 * 1024.0 * 1024.0));
                else
                    std::fprintf(stderr, "Info: Using GPU %d: %s\n", cfg.cuda_device_id, name.c_str());
            }
        }
        if (cfg.use_cuda_streams) {
            s = detail::hip_status(pcr_hip_stream_create(&stream));
            if (!s.ok()) return s;
            own_stream = true;
        }

        // accumulation groups + outputs
        for (const auto& r : cfg.reductions) {
            int gi = -1;
            for (size_t k = 0; k < groups.size(); ++k)
                if (groups[k].value_channel == r.value_channel && same_glyph(groups[k].glyph, r.glyph)) gi = (int)k;
            if (gi < 0) {
                groups.emplace_back();
                gi = (int)groups.size() - 1;
                groups[gi].value_channel = r.value_channel;
                groups[gi].glyph = r.glyph;
            }
            // a glyph reduction of an unsupported type is rejected at ingest (as in the reference);
            // it still gets planes so that finalize has something to read.
            groups[gi].mask |= planes_for(r.type);
            Output o;
            o.group = gi;
            o.type = r.type;
            o.band_name = r.output_band_name.empty()
                ? r.value_channel + "_" + std::to_string(static_cast<int>(r.type)) : r.output_band_name;
            outputs.push_back(o);
        }

        // row window of this device
        int r0 = 0, r1 = g.height;
        if (cfg.shard_row_begin >= 0 || cfg.shard_row_end >= 0) {
            r0 = cfg.shard_row_begin < 0 ? 0 : cfg.shard_row_begin;
            r1 = cfg.shard_row_end < 0 ? g.height : cfg.shard_row_end;
            if (r0 > r1 || r1 > g.height)
                return Status::error(StatusCode::InvalidArgument, "pipeline: shard row range outside the grid");
            for (const auto& gr : groups) halo = std::max(halo, reach_rows(gr.glyph));
            halo = std::max(halo, cfg.shard_halo_rows);
            halo = std::min(halo, g.height);
            // -2 (the out-of-core bands): the block is made of whole reference-tile rows, footprints are clipped to the tile
            // of their centre cell (Q4), nothing can land outside the block -- no apron rows at all
            if (cfg.shard_halo_rows == -2 && r0 % g.tile_height == 0 && (r1 % g.tile_height == 0 || r1 == g.height)) halo = 0;
        }
        hg.min_x = g.bounds.min_x; hg.min_y = g.bounds.min_y; hg.max_x = g.bounds.max_x; hg.max_y = g.bounds.max_y;
        hg.cell_size_x = g.cell_size_x; hg.cell_size_y = g.cell_size_y;
        hg.width = g.width; hg.height = g.height;
        hg.tile_width = g.tile_width; hg.tile_height = g.tile_height;
        hg.own_row0 = r0; hg.own_row1 = r1;
        hg.state_row0 = std::max(0, r0 - halo);
        hg.state_rows = std::min(g.height, r1 + halo) - hg.state_row0;

        {
            // arena pre-sized at create, like the reference's MemoryPool; a pool that cannot be created follows the
            // reference's matrix too (src/engine/pipeline.cpp:181-192)
            const int rc = pcr_hip_engine_create(&engine, &hg, cfg.gpu_pool_size_bytes, stream);
            if (rc == PCR_HIP_OUT_OF_MEMORY) {
                const std::string err_msg = "Failed to create GPU memory pool";
                if (cfg.gpu_require_strict || !cfg.gpu_fallback_to_cpu) return Status::error(StatusCode::OutOfMemory, err_msg);
                std::fprintf(stderr, "Warning: %s - falling back to CPU mode\n", err_msg.c_str());
                return no_cpu_engine("the reference would fall back to its CPU mode here");
            }
            s = detail::hip_status(rc);
            if (!s.ok()) return s;
        }
        s = detail::hip_status(pcr_hip_engine_set_path(engine, cfg.scatter_path));
        if (!s.ok()) return s;

        const int64_t cells = (int64_t)hg.state_rows * hg.width;
        for (auto& gr : groups) {
            for (int p = 0; p < 4; ++p) {
                if (!(gr.mask & kPlaneBits[p])) continue;
                s = gr.planes[p].allocate((size_t)std::max<int64_t>(cells, 1) * sizeof(float), MemoryLocation::Device);
                if (!s.ok()) return s;
            }
            gr.view.d_sum = static_cast<float*>(gr.planes[0].data());
            gr.view.d_wgt = static_cast<float*>(gr.planes[1].data());
            gr.view.d_max = static_cast<float*>(gr.planes[2].data());
            gr.view.d_min = static_cast<float*>(gr.planes[3].data());
        }
        if (!outputs.empty() && !(s = allocate_result()).ok()) return s;
        if (!(s = d_bands_done.allocate(std::max<size_t>(groups.size(), 1) * sizeof(uint32_t), MemoryLocation::Device)).ok()) return s;
        return detail::hip_status(pcr_hip_stream_synchronize(stream));
    }

    // Identity values into a group's planes, for every reader that may come before the group's first scatter (finalize
    // of an empty pipeline, checkpoints, the halo exchange of a rank without points, state_planes()).
    Status define_planes(Group& gr) {
        if (gr.defined) return Status::success();
        const int64_t cells = (int64_t)hg.state_rows * hg.width;
        for (int p = 0; p < 4; ++p) {
            if (!(gr.mask & kPlaneBits[p])) continue;
            const float ident = p == 2 ? -3.402823466e+38f : p == 3 ? 3.402823466e+38f : 0.0f;
            Status s = detail::hip_status(pcr_hip_plane_fill(static_cast<float*>(gr.planes[p].data()), ident, cells, stream));
            if (!s.ok()) return s;
        }
        gr.defined = true;
        return Status::success();
    }
    Status define_all_planes() {
        for (auto& gr : groups) {
            Status s = define_planes(gr);
            if (!s.ok()) return s;
        }
        return Status::success();
    }

    // rows a glyph can reach above/below its centre row (sizes the halo of a row-block shard)
    int reach_rows(const GlyphSpec& gl) const {
        double cap = std::min<double>(std::max(0.0f, gl.max_radius_cells), 1 << 20);
        if (gl.type == GlyphType::Gaussian) return (int)std::ceil(cap);
        if (gl.type == GlyphType::Line) {
            double hy = std::fabs((double)gl.default_half_length / cfg.grid.cell_size_y);
            return (int)std::ceil(std::max(hy, cap)) + 1;
        }
        return 0;
    }

    // A footprint is clipped to the reference tile of its centre cell (Q4): when the owned block is made of whole
    // tile rows nothing can land outside it, whatever the glyph.
    bool block_is_whole_tiles() const {
        const int th = cfg.grid.tile_height;
        return hg.own_row0 % th == 0 && (hg.own_row1 % th == 0 || hg.own_row1 == hg.height);
    }

    // Row-block shards: a Line glyph with a per-point half_length channel reaches |hl_i / cell_size_y| rows (+1 for
    // the rounding of the end points), and on north-up grids max_radius_cells does not cap that (hy < 0 passes
    // std::min(h, cap) untouched, glyph_kernels.cu:228-234).  Cells beyond the halo would be clipped by the state
    // window and never reach their owner: refuse instead of returning a grid that differs from the unsharded one.
    // rows_needed: the reach of the longest segment among the points the filter keeps (0: the check does not apply).
    Status line_reach_rows(const GlyphSpec& gl, const void* d_half_length, const uint8_t* d_mask, size_t n, int* rows_needed) {
        *rows_needed = 0;
        if (gl.type != GlyphType::Line || !d_half_length) return Status::success();
        if (own_rows() == hg.height || block_is_whole_tiles()) return Status::success();
        detail::Buffer& word = staging["line_reach:word"];           // the reduction's device word, kept for the pipeline's life
        if (word.bytes() < 4) {
            Status a = word.allocate(256, MemoryLocation::Device);
            if (!a.ok()) return a;
        }
        // hy = half_length / cell_size_y goes through std::min(hy, cap) (glyph_kernels.cu:228-234): the sign of half_length
        // that makes hy positive is capped by max_radius_cells, the other one reaches |half_length / cell_size_y| rows
        float max_pos = 0.f, max_neg = 0.f;
        Status s = detail::hip_status(pcr_hip_signed_max_f32_masked(static_cast<const float*>(d_half_length), d_mask, n,
                                                                    static_cast<uint32_t*>(word.data()), &max_pos, &max_neg, stream));
        if (!s.ok()) return s;
        const double acsy = std::fabs(cfg.grid.cell_size_y);
        const float capped_side = cfg.grid.cell_size_y > 0 ? max_pos : max_neg;
        const float free_side = cfg.grid.cell_size_y > 0 ? max_neg : max_pos;
        double rows = std::max(std::ceil((double)free_side / acsy),
                               std::min(std::ceil((double)capped_side / acsy), std::ceil((double)std::max(gl.max_radius_cells, 0.0f))));
        rows += 1.0;                                                 // rounding of the end points
        rows = std::min<double>(rows, cfg.grid.tile_height - 1);
        *rows_needed = (int)rows;
        return Status::success();
    }

    Status reach_error(int rows) const {
        return Status::error(StatusCode::InvalidArgument,
            "pipeline: a Line segment of this cloud reaches " + std::to_string((long long)rows) +
            " rows beyond its centre row, but this row-block shard keeps a halo of " + std::to_string(halo) +
            " rows (sized from default_half_length / max_radius_cells); set PipelineConfig.shard_halo_rows >= " +
            std::to_string((long long)rows) + " on every rank, or use tile-aligned row blocks");
    }

    // The pipeline's device is made current for the duration of a call that launches or allocates, whatever the
    // calling thread had current (torch, another pipeline).
    struct DeviceScope {
        int prev = -1;
        bool changed = false;
        explicit DeviceScope(int dev) {
            if (pcr_hip_get_device(&prev) == PCR_HIP_OK && prev != dev) changed = pcr_hip_set_device(dev) == PCR_HIP_OK;
        }
        ~DeviceScope() { if (changed) pcr_hip_set_device(prev); }
        DeviceScope(const DeviceScope&) = delete;
        DeviceScope& operator=(const DeviceScope&) = delete;
    };

    // Device pointer of a named array of the cloud; host-resident arrays are staged to HBM.
    Status device_array(const void* src, MemoryLocation loc, size_t bytes, const std::string& key,
                        const void** out) {
        if (loc == MemoryLocation::Device) { *out = src; return Status::success(); }
        detail::Buffer& b = staging[key];
        if (b.bytes() < bytes) {
            // earlier kernels may still read the old block
            Status s = detail::hip_status(pcr_hip_stream_synchronize(stream));
            if (!s.ok()) return s;
            s = b.allocate(bytes + bytes / 8, MemoryLocation::Device);
            if (!s.ok()) return s;
        }
        Status s = detail::hip_status(pcr_hip_memcpy_h2d(b.data(), src, bytes, stream));
        if (!s.ok()) return s;
        *out = b.data();
        return Status::success();
    }

    // Rows the Line groups of this pipeline need beyond a centre row for THIS cloud (0 when no check applies): what
    // ingest() would refuse above the shard's halo.  A sharded caller reduces it over the ranks (MAX) first, so that
    // every rank refuses together instead of one rank raising while the others wait in a collective.  The filter is
    // not applied here (an upper bound: ingest() itself checks the kept points only).
    Status query_line_reach(const PointCloud& cloud, int* rows_out) {
        *rows_out = 0;
        const size_t n = cloud.count();
        if (n == 0) return Status::success();
        DeviceScope dev(cfg.cuda_device_id);
        for (const auto& gr : groups) {
            if (gr.glyph.type != GlyphType::Line || gr.glyph.half_length_channel.empty()) continue;
            const ChannelDesc* d = cloud.channel(gr.glyph.half_length_channel);
            if (!d || d->dtype != DataType::Float32) continue;
            const void* hl = nullptr;
            Status s = device_array(cloud.channel_data(gr.glyph.half_length_channel), cloud.location(), n * sizeof(float),
                                    "ch:" + gr.glyph.half_length_channel, &hl);
            if (!s.ok()) return s;
            int rows = 0;
            if (!(s = line_reach_rows(gr.glyph, hl, nullptr, n, &rows)).ok()) return s;
            *rows_out = std::max(*rows_out, rows);
        }
        return Status::success();
    }

    Status ingest(const PointCloud& cloud, bool wait = true) {
        const size_t n = cloud.count();
        if (n == 0) return Status::success();
        DeviceScope dev(cfg.cuda_device_id);
        // validate every predicate and reduction before touching state (pipeline_common.cpp: the reference's checks and messages)
        {
            Status ok = detail::validate_cloud(cfg, cloud, PCR_HIP_MAX_FILTER_SET, PCR_HIP_MAX_FILTER_PREDICATES);
            if (!ok.ok()) return ok;
        }

        const MemoryLocation loc = cloud.location();
        const void *dx = nullptr, *dy = nullptr;
        Status s = device_array(cloud.x(), loc, n * sizeof(double), "x", &dx);
        if (!s.ok()) return s;
        s = device_array(cloud.y(), loc, n * sizeof(double), "y", &dy);
        if (!s.ok()) return s;

        std::map<std::string, const void*> staged;      // a channel is staged once per ingest, whoever asks first
        auto f32_channel = [&](const std::string& name, const void** out) -> Status {
            *out = nullptr;
            if (name.empty()) return Status::success();
            const ChannelDesc* d = cloud.channel(name);
            if (!d || d->dtype != DataType::Float32) return Status::success();   // -> GlyphSpec default
            auto hit = staged.find(name);
            if (hit != staged.end()) { *out = hit->second; return Status::success(); }
            Status st = device_array(cloud.channel_data(name), loc, n * sizeof(float), "ch:" + name, out);
            if (st.ok()) staged[name] = *out;
            return st;
        };

        // Filter stage, on the device: a byte mask evaluated once per ingest and honoured by every
        // routing kernel.  (The reference gathers values by filter index but routes the UNFILTERED
        // cloud, pipeline.cpp:436-438 vs :662, which is why its own WithFilter test is disabled;
        // here a filtered-out point simply does not exist for any reduction.)
        size_t kept = n;
        const uint8_t* active_mask = nullptr;
        if (!cfg.filter.empty()) {
            std::vector<pcr_hip_predicate> preds(cfg.filter.predicates.size());
            for (size_t k = 0; k < preds.size(); ++k) {
                const FilterPredicate& pr = cfg.filter.predicates[k];
                const void* ch = nullptr;
                if (!(s = f32_channel(pr.channel_name, &ch)).ok()) return s;
                preds[k].d_channel = static_cast<const float*>(ch);
                preds[k].op = static_cast<int32_t>(pr.op);
                preds[k].value = pr.value;
                preds[k].set_size = static_cast<int32_t>(pr.value_set.size());
                for (size_t j = 0; j < pr.value_set.size(); ++j) preds[k].set[j] = pr.value_set[j];
            }
            detail::Buffer& mb = staging["filter:mask"];
            if (mb.bytes() < n + 8) {
                if (!(s = detail::hip_status(pcr_hip_stream_synchronize(stream))).ok()) return s;
                if (!(s = mb.allocate(n + n / 8 + 16, MemoryLocation::Device)).ok()) return s;
            }
            // layout: [u64 survivor count][mask bytes]
            auto* d_count = static_cast<unsigned long long*>(mb.data());
            auto* d_mask = static_cast<uint8_t*>(mb.data()) + 8;
            s = detail::hip_status(pcr_hip_filter_mask(preds.data(), (int)preds.size(), n, d_mask, d_count, stream));
            if (!s.ok()) return s;
            unsigned long long h_count = 0;
            if (!(s = detail::hip_status(pcr_hip_memcpy_d2h(&h_count, d_count, sizeof h_count, stream))).ok()) return s;
            if (!(s = detail::hip_status(pcr_hip_stream_synchronize(stream))).ok()) return s;
            kept = (size_t)h_count;
            if (kept == 0) return Status::success();             // pipeline.cpp:349-353
            pcr_hip_engine_set_point_mask(engine, d_mask);
            active_mask = d_mask;
        }
        struct MaskGuard {
            pcr_hip_engine* e;
            ~MaskGuard() { pcr_hip_engine_set_point_mask(e, nullptr); }
        } mask_guard{engine};

        // Row-block shards: every Line group's reach is checked BEFORE the first scatter of this ingest, so a refused
        // cloud leaves no group half-accumulated (and, sharded, every rank can agree on the verdict first:
        // Pipeline::line_reach_rows + pcr.distributed.ShardedPipeline.ingest).
        int reach_needed = 0;
        for (const auto& gr : groups) {
            if (gr.glyph.type != GlyphType::Line) continue;
            const void* hl = nullptr;
            if (!(s = f32_channel(gr.glyph.half_length_channel, &hl)).ok()) return s;
            int rows = 0;
            if (!(s = line_reach_rows(gr.glyph, hl, active_mask, n, &rows)).ok()) return s;
            reach_needed = std::max(reach_needed, rows);
        }
        if (reach_needed > halo) return reach_error(reach_needed);

        bands_stale();                                   // (this cloud's points change the planes and the touched flags)
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            Group& gr = groups[gi];
            const void* dv = nullptr;
            s = f32_channel(gr.value_channel, &dv);
            if (!s.ok()) return s;
            // 2: the planes are still undefined -- the scatter defines every cell of the state window itself (the binned
            // Point path at no extra pass, every other path by filling first); 1: identity-filled, nothing accumulated yet
            pcr_hip_engine_planes_fresh(engine, !gr.defined ? 2 : (gr.fresh && gr.glyph.type == GlyphType::Point ? 1 : 0));
            if (gr.glyph.type == GlyphType::Point) {
                gr.fresh = false;
                const bool offered = offer_bands(gi);
                s = detail::hip_status(pcr_hip_scatter_point(
                    engine, gr.mask, &gr.view, static_cast<const double*>(dx), static_cast<const double*>(dy),
                    static_cast<const float*>(dv), n));
                gr.bands_with_scatter = offered && s.ok() && pcr_hip_engine_finalize_taken(engine) == 1;
            } else {
                pcr_hip_glyph hgph{};
                hgph.type = static_cast<int32_t>(gr.glyph.type);
                hgph.default_direction = gr.glyph.default_direction;
                hgph.default_half_length = gr.glyph.default_half_length;
                hgph.default_sigma_x = gr.glyph.default_sigma_x;
                hgph.default_sigma_y = gr.glyph.default_sigma_y;
                hgph.default_rotation = gr.glyph.default_rotation;
                hgph.max_radius_cells = gr.glyph.max_radius_cells;
                const void* p = nullptr;
                if (!(s = f32_channel(gr.glyph.direction_channel, &p)).ok()) return s;
                hgph.d_direction = static_cast<const float*>(p);
                if (!(s = f32_channel(gr.glyph.half_length_channel, &p)).ok()) return s;
                hgph.d_half_length = static_cast<const float*>(p);
                if (!(s = f32_channel(gr.glyph.sigma_x_channel, &p)).ok()) return s;
                hgph.d_sigma_x = static_cast<const float*>(p);
                if (!(s = f32_channel(gr.glyph.sigma_y_channel, &p)).ok()) return s;
                hgph.d_sigma_y = static_cast<const float*>(p);
                if (!(s = f32_channel(gr.glyph.rotation_channel, &p)).ok()) return s;
                hgph.d_rotation = static_cast<const float*>(p);
                s = detail::hip_status(pcr_hip_scatter_glyph(
                    engine, &hgph, gr.mask & (PCR_HIP_PLANE_SUM | PCR_HIP_PLANE_WGT), &gr.view,
                    static_cast<const double*>(dx), static_cast<const double*>(dy),
                    static_cast<const float*>(dv), n));
            }
            gr.fresh = false;
            if (!s.ok()) return s;
            gr.defined = true;               // (n > 0 here: the scatter ran)
        }
        // host arrays may be reused by the caller as soon as we return (ingest_async: page-locked arrays are
        // read by the DMA engine later, the caller keeps them alive until synchronize())
        if (loc != MemoryLocation::Device && (wait || loc != MemoryLocation::HostPinned)) {
            s = detail::hip_status(pcr_hip_stream_synchronize(stream));
            if (!s.ok()) return s;
        }

        points += kept;                 // points_processed += filtered_count (pipeline.cpp:749)
        collections++;
        if (callback) {
            ProgressInfo info = stats();
            if (!callback(info))
                return Status::error(StatusCode::InvalidArgument, "pipeline: cancelled by user");
        }
        return Status::success();
    }

    // The result grid (and, for a host-resident result, its device-side band buffers) is allocated
    // once, at create: finalize only launches kernels.
    Status allocate_result() {
        const int rows = own_rows();
        const int W = hg.width;
        std::vector<BandDesc> bands;
        for (const auto& o : outputs) {
            BandDesc b;
            b.name = o.band_name;
            b.dtype = DataType::Float32;
            b.is_state = false;
            bands.push_back(b);
        }
        if (bands.empty()) return Status::error(StatusCode::OutOfMemory, "pipeline: failed to allocate result grid");
        if (rows <= 0) { result.reset(); return Status::success(); }
        const bool on_device = cfg.result_location == MemoryLocation::Device;
        result = on_device ? Grid::create(W, rows, bands, MemoryLocation::Device)
                           : Grid::create_host_page_locked(W, rows, bands);
        if (!result) return Status::error(StatusCode::OutOfMemory, "pipeline: failed to allocate result grid");
        if (!on_device) {
            d_bands.resize(outputs.size());
            for (auto& b : d_bands) {
                Status s = b.allocate((size_t)rows * W * sizeof(float), MemoryLocation::Device);
                if (!s.ok()) return s;
            }
        }
        return Status::success();
    }

    Status finalize(bool wait = true) {
        const int rows = own_rows();
        const int W = hg.width;
        if (outputs.empty()) return Status::error(StatusCode::OutOfMemory, "pipeline: failed to allocate result grid");
        if (rows <= 0) return Status::success();
        DeviceScope dev(cfg.cuda_device_id);
        const bool on_device = cfg.result_location == MemoryLocation::Device;
        if (!result) {
            Status as = allocate_result();
            if (!as.ok()) return as;
        }
        {
            Status ds = define_all_planes();          // a group no cloud ever reached
            if (!ds.ok()) return ds;
        }
        uint32_t* d_touched = nullptr;
        Status s = detail::hip_status(pcr_hip_engine_tile_touched(engine, &d_touched, nullptr, nullptr));
        if (!s.ok()) return s;
        // one sweep per accumulation group: its planes are read once for all of its bands
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            std::vector<int> types;
            std::vector<float*> dsts;
            std::vector<size_t> bands_of;
            auto flush = [&]() -> Status {
                if (types.empty()) return Status::success();
                // (a group whose bands the defining scatter stored: the kernel returns at once when the device word says so)
                const uint32_t* done = groups[gi].bands_with_scatter ? static_cast<const uint32_t*>(d_bands_done.data()) + gi : nullptr;
                Status fs = detail::hip_status(pcr_hip_finalize_group_unless(&hg, &groups[gi].view, d_touched, (int)types.size(),
                                                                             types.data(), dsts.data(), done, stream));
                types.clear();
                dsts.clear();
                return fs;
            };
            for (size_t r = 0; r < outputs.size(); ++r) {
                if (outputs[r].group != (int)gi) continue;
                types.push_back(static_cast<int>(outputs[r].type));
                dsts.push_back(band_device(r));
                bands_of.push_back(r);
                if (types.size() == PCR_HIP_MAX_FINALIZE_OUTPUTS && !(s = flush()).ok()) return s;
            }
            if (!(s = flush()).ok()) return s;
            if (!on_device) {
                for (size_t r : bands_of) {
                    s = detail::hip_status(pcr_hip_memcpy_d2h(result->band_f32((int)r), d_bands[r].data(),
                                                              (size_t)rows * W * sizeof(float), stream));
                    if (!s.ok()) return s;
                }
            }
        }
        // finalize_async: a device-resident result is stream-ordered like everything else on the device
        if (wait || !on_device || !cfg.output_path.empty()) {
            s = detail::hip_status(pcr_hip_stream_synchronize(stream));
            if (!s.ok()) return s;
        }
        finalized = true;
        if (!cfg.output_path.empty()) {
            // pipeline.cpp:1351-1361 of the reference: the finalized grid goes to output_path as GeoTIFF.
            // A shard writes its own row block (georeferenced as such); a device-resident result is copied out first.
            GridConfig out_cfg = cfg.grid;
            if (rows != cfg.grid.height) {
                out_cfg.height = rows;
                out_cfg.bounds.max_y = cfg.grid.bounds.max_y + hg.own_row0 * cfg.grid.cell_size_y;
                out_cfg.bounds.min_y = out_cfg.bounds.max_y + rows * cfg.grid.cell_size_y;
                if (out_cfg.bounds.min_y > out_cfg.bounds.max_y) std::swap(out_cfg.bounds.min_y, out_cfg.bounds.max_y);
            }
            if (on_device) {
                std::unique_ptr<Grid> host = result->to(MemoryLocation::Host);
                if (!host) return Status::error(StatusCode::OutOfMemory, "pipeline: failed to copy the result grid to the host");
                return write_geotiff(cfg.output_path, *host, out_cfg, GeoTiffOptions());
            }
            return write_geotiff(cfg.output_path, *result, out_cfg, GeoTiffOptions());
        }
        return Status::success();
    }

    // ---- `.pcrt` checkpoints ---------------------------------------------------------------
    // A whole-grid pipeline checkpoints every touched tile.  A row-block shard WRITES the tiles it owns whole -- which is all
    // of them when its block is made of whole reference-tile rows (ShardedPipeline(align = tile_height); the N = 2 and 4 blocks
    // of C5): every rank writes its own files into the same directory and their union is the pipeline's checkpoint, readable
    // by an unsharded pipeline too.  A block that cuts tiles shares them with its neighbours: ShardedPipeline::save_state
    // gathers the state to one rank for those.  Any shard LOADS: it takes its own rows out of every tile file they meet.
    Status checkpoint_dir(const std::string& dir_in, std::string* dir, bool writing) const {
        *dir = dir_in.empty() ? cfg.state_dir : dir_in;
        if (dir->empty()) return Status::error(StatusCode::InvalidArgument, "pipeline: no state directory given");
        if (writing && own_rows() != hg.height && !block_is_whole_tiles())
            return Status::error(StatusCode::NotImplemented,
                "pipeline: this row block cuts a reference tile it shares with a neighbour, so no rank can write that tile alone: "
                "ShardedPipeline.save_state gathers the state for it, or shard with align = tile_height");
        return Status::success();
    }
    detail::StateWindow state_window(std::vector<std::vector<float>>& planes) const {
        detail::StateWindow w;
        w.row0 = hg.state_row0;
        w.rows = hg.state_rows;
        w.own_row0 = hg.own_row0;
        w.own_row1 = hg.own_row1;
        w.plane = [&planes](int g, int p) -> float* {
            auto& v = planes[(size_t)g * 4 + (size_t)p];
            return v.empty() ? nullptr : v.data();
        };
        return w;
    }

    // The state WINDOW of this pipeline (any shard) as host copies: planes[4 g + p] (state_rows x W floats; empty when group g
    // has no plane p) + the touched flags.  What the out-of-core driver parks between two visits of a band.
    Status export_window(std::vector<std::vector<float>>& planes, std::vector<uint32_t>& touched) {
        DeviceScope dev(cfg.cuda_device_id);
        Status s = define_all_planes();
        if (!s.ok()) return s;
        const size_t cells = (size_t)hg.state_rows * hg.width;
        planes.assign(groups.size() * 4, {});
        for (size_t gi = 0; gi < groups.size(); ++gi)
            for (int p = 0; p < 4; ++p) {
                if (!(groups[gi].mask & kPlaneBits[p])) continue;
                planes[gi * 4 + p].resize(cells);
                s = detail::hip_status(pcr_hip_memcpy_d2h(planes[gi * 4 + p].data(), groups[gi].planes[p].data(), cells * sizeof(float), stream));
                if (!s.ok()) return s;
            }
        uint32_t* d_touched = nullptr;
        int tx = 0, ty = 0;
        if (!(s = detail::hip_status(pcr_hip_engine_tile_touched(engine, &d_touched, &tx, &ty))).ok()) return s;
        touched.resize((size_t)tx * ty);
        if (!(s = detail::hip_status(pcr_hip_memcpy_d2h(touched.data(), d_touched, touched.size() * 4, stream))).ok()) return s;
        return detail::hip_status(pcr_hip_stream_synchronize(stream));
    }
    Status import_window(const std::vector<std::vector<float>>& planes, const std::vector<uint32_t>& touched) {
        bands_stale();
        DeviceScope dev(cfg.cuda_device_id);
        const size_t cells = (size_t)hg.state_rows * hg.width;
        if (planes.size() != groups.size() * 4)
            return Status::error(StatusCode::InvalidArgument, "pipeline: parked state does not match the pipeline's groups");
        Status s = Status::success();
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            for (int p = 0; p < 4; ++p) {
                if (!(groups[gi].mask & kPlaneBits[p])) continue;
                if (planes[gi * 4 + p].size() != cells)
                    return Status::error(StatusCode::InvalidArgument, "pipeline: parked state does not match the band's window");
                s = detail::hip_status(pcr_hip_memcpy_h2d(groups[gi].planes[p].data(), planes[gi * 4 + p].data(), cells * sizeof(float), stream));
                if (!s.ok()) return s;
            }
            groups[gi].defined = true;
            groups[gi].fresh = false;
        }
        uint32_t* d_touched = nullptr;
        int tx = 0, ty = 0;
        if (!(s = detail::hip_status(pcr_hip_engine_tile_touched(engine, &d_touched, &tx, &ty))).ok()) return s;
        if (touched.size() != (size_t)tx * ty)
            return Status::error(StatusCode::InvalidArgument, "pipeline: parked touched flags do not match the tile grid");
        if (!(s = detail::hip_status(pcr_hip_memcpy_h2d(d_touched, touched.data(), touched.size() * 4, stream))).ok()) return s;
        return detail::hip_status(pcr_hip_stream_synchronize(stream));
    }

    std::vector<detail::StateOutput> state_outputs() const {
        std::vector<detail::StateOutput> o;
        for (const auto& out : outputs) o.push_back({out.group, out.type});
        return o;
    }

    Status save_state(const std::string& dir_in) {
        std::string dir;
        Status s = checkpoint_dir(dir_in, &dir, true);
        if (!s.ok()) return s;
        std::vector<std::vector<float>> planes;
        std::vector<uint32_t> touched;
        if (!(s = export_window(planes, touched)).ok()) return s;
        return detail::write_state_tiles(cfg.grid, state_outputs(), state_window(planes), touched, dir);
    }

    Status load_state(const std::string& dir_in) {
        std::string dir;
        Status s = checkpoint_dir(dir_in, &dir, false);
        if (!s.ok()) return s;
        std::vector<std::vector<float>> planes;
        std::vector<uint32_t> touched;
        if (!(s = export_window(planes, touched)).ok()) return s;          // files overlay the current state
        size_t loaded = 0;
        if (!(s = detail::read_state_tiles(cfg.grid, state_outputs(), state_window(planes), touched, dir, &loaded)).ok()) return s;
        if (!loaded) return Status::success();
        return import_window(planes, touched);                              // (drops bands a scatter stored, marks the planes defined)
    }

    ProgressInfo stats() const {
        ProgressInfo info;
        info.collections_processed = collections;
        info.collections_total = 0;
        info.points_processed = points;
        info.tiles_active = 0;
        DeviceScope dev(cfg.cuda_device_id);
        uint32_t* d_touched = nullptr;
        int tx = 0, ty = 0;
        if (engine && pcr_hip_engine_tile_touched(engine, &d_touched, &tx, &ty) == PCR_HIP_OK) {
            std::vector<uint32_t> h((size_t)tx * ty);
            if (pcr_hip_memcpy_d2h(h.data(), d_touched, h.size() * sizeof(uint32_t), stream) == PCR_HIP_OK &&
                pcr_hip_stream_synchronize(stream) == PCR_HIP_OK)
                for (uint32_t v : h) info.tiles_active += v ? 1 : 0;
        }
        info.elapsed_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
        return info;
    }
};



// ---- out-of-core grids: row bands of whole reference-tile rows ---------------------------------------------------
// The reference keeps every tile's state behind a TileManager: an LRU cache in memory, evicted tiles flushed to `.pcrt`
// files and loaded back on the next acquire (src/engine/tile_manager.cpp:76-138, 183-375), so that a grid may be larger
// than memory.  This build keeps the state of the WHOLE grid in HBM -- until it does not fit the budget.
// Then the grid is swept in bands of whole reference-tile rows: footprints are clipped to the reference tile of their
// centre cell (Q4), so nothing a band's points paint can land outside the band -- a band is an ordinary row-block shard
// with no halo to exchange.  One band's planes are in HBM at a time (a sub-pipeline created for the visit); the others are
// parked as host copies up to host_cache_budget and, least recently used first, in files under state_dir beyond it.  Every
// ingest visits every band (the kernels keep the points whose centre row the band owns); finalize visits them once more
// and assembles the host result.  Results are those of the in-core pipeline bit for bit: the same kernels run on the same
// points of each tile, in the same order.
struct Pipeline::Banded {
    PipelineConfig cfg;                                   // the WHOLE grid, as the caller gave it
    std::vector<std::pair<int, int>> bands;               // [r0, r1), multiples of the tile height
    detail::Grouping grouping;                            // groups' plane masks, one StateOutput per ReductionSpec
    struct Parked {
        bool any = false, on_disk = false;
        std::vector<std::vector<float>> planes;           // [4 g + p]: the band's window of plane p of group g (empty: no such plane)
        std::vector<uint32_t> touched;                    // tiles_x * tiles_y flags of the whole grid (only this band's rows are set)
        size_t bytes = 0;
        uint64_t stamp = 0;
    };
    std::vector<Parked> parked;
    size_t host_budget = 0, host_used = 0, spills = 0, reloads = 0;
    uint64_t clock = 0;
    // Evicted bands live in the reference's own format and layout: one `.pcrt` file per touched reference tile and
    // ReductionSpec (tile_RRRR_CCCC.pcrt; reduction_<i>/ for several reductions) -- what the reference's TileManager flushes
    // on eviction (src/engine/tile_manager.cpp:76-138 -> src/io/tile_state_io.cpp:45-95) -- in a directory of the pipeline's
    // own (under state_dir, else the temporary directory), removed with the pipeline: a spill is working state, possibly
    // partial and older than a band's host copy, and must never be mistaken for a checkpoint.  save_state() writes one.
    std::string spill_dir;
    std::unique_ptr<Grid> result;
    bool finalized = false;
    size_t collections = 0, points = 0, tiles_active = 0;
    ProgressCallback callback;
    ScatterInfo last{};
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();

    ~Banded() {
        if (!spill_dir.empty()) {
            std::error_code ec;
            std::filesystem::remove_all(spill_dir, ec);
        }
    }

    // bytes of device memory one grid row costs: 4 B per cell and plane, + the finalized bands a sub-pipeline keeps
    static size_t bytes_per_row(const PipelineConfig& c) {
        const detail::Grouping g = detail::group_reductions(c.reductions);
        size_t planes = 0;
        for (uint32_t m : g.masks)
            for (int p = 0; p < 4; ++p) planes += (m & kPlaneBits[p]) ? 1 : 0;
        return (size_t)c.grid.width * 4 * (planes + c.reductions.size());
    }

    detail::StateWindow window_of(size_t b, Parked& k) const {
        detail::StateWindow w;
        w.row0 = bands[b].first;
        w.rows = bands[b].second - bands[b].first;
        w.plane = [&k](int g, int p) -> float* {
            auto& v = k.planes[(size_t)g * 4 + (size_t)p];
            return v.empty() ? nullptr : v.data();
        };
        return w;
    }

    // a band nobody has touched yet: identity planes, no flags
    void blank(size_t b, Parked& k) const {
        const size_t cells = (size_t)(bands[b].second - bands[b].first) * (size_t)cfg.grid.width;
        k.planes.assign(grouping.masks.size() * 4, {});
        for (size_t g = 0; g < grouping.masks.size(); ++g)
            for (int p = 0; p < 4; ++p)
                if (grouping.masks[g] & kPlaneBits[p])
                    k.planes[g * 4 + (size_t)p].assign(cells, p == 2 ? -3.402823466e+38f : p == 3 ? 3.402823466e+38f : 0.0f);
        const GridConfig& g = cfg.grid;
        k.touched.assign((size_t)((g.width + g.tile_width - 1) / g.tile_width) * (size_t)((g.height + g.tile_height - 1) / g.tile_height), 0u);
    }

    Status spill(size_t b) {
        Parked& k = parked[b];
        Status s = detail::write_state_tiles(cfg.grid, grouping.outputs, window_of(b, k), k.touched, spill_dir);
        if (!s.ok()) return s;
        host_used -= k.bytes;
        std::vector<std::vector<float>>().swap(k.planes);
        std::vector<uint32_t>().swap(k.touched);
        k.on_disk = true;
        ++spills;
        return Status::success();
    }

    Status reload(size_t b) {
        Parked& k = parked[b];
        blank(b, k);
        size_t loaded = 0;
        Status s = detail::read_state_tiles(cfg.grid, grouping.outputs, window_of(b, k), k.touched, spill_dir, &loaded);
        if (!s.ok()) return s;
        k.on_disk = false;
        host_used += k.bytes;
        ++reloads;
        return Status::success();
    }

    // the host copy of a band that is unchanged since it was written to disk is dropped again (finalize, save_state)
    void drop_host_copy(size_t b) {
        Parked& k = parked[b];
        host_used -= k.bytes;
        std::vector<std::vector<float>>().swap(k.planes);
        std::vector<uint32_t>().swap(k.touched);
        k.on_disk = true;
    }

    // park band b's state (already in k.planes / k.touched) and evict the least recently used bands beyond the host budget
    Status account(size_t b) {
        Parked& k = parked[b];
        size_t bytes = k.touched.size() * 4;
        for (const auto& pl : k.planes) bytes += pl.size() * 4;
        host_used += bytes - (k.any && !k.on_disk ? k.bytes : 0);
        k.bytes = bytes;
        k.any = true;
        k.on_disk = false;
        k.stamp = ++clock;
        return evict();
    }
    Status evict() {
        while (host_used > host_budget) {
            size_t victim = parked.size();
            for (size_t i = 0; i < parked.size(); ++i)
                if (parked[i].any && !parked[i].on_disk && (victim == parked.size() || parked[i].stamp < parked[victim].stamp)) victim = i;
            if (victim == parked.size()) break;
            Status s = spill(victim);
            if (!s.ok()) return s;
        }
        return Status::success();
    }

    // was_on_disk: the band's state came from its files and those are still its current state
    std::unique_ptr<Pipeline> visit(size_t b, Status* st, bool* was_on_disk = nullptr) {
        PipelineConfig c = cfg;
        c.shard_row_begin = bands[b].first;
        c.shard_row_end = bands[b].second;
        c.result_location = MemoryLocation::Host;
        c.finalize_with_first_ingest = false;             // (a band's visit ends with its state parked, never with finalize)
        c.shard_halo_rows = -2;                           // whole tile rows: the band's state window is the band
        c.output_path.clear();
        c.state_dir.clear();
        c.resume = false;
        c.exec_mode = ExecutionMode::GPU;
        c.gpu_fallback_to_cpu = false;                    // (a band is a device pipeline or an error)
        if (was_on_disk) *was_on_disk = false;
        std::unique_ptr<Pipeline> sub = Pipeline::create(c);
        if (!sub) { *st = Status::error(StatusCode::OutOfMemory, "pipeline: out-of-core band could not be created: " + pipeline_create_error()); return nullptr; }
        Parked& k = parked[b];
        if (k.any) {
            if (k.on_disk) {
                if (!(*st = reload(b)).ok()) return nullptr;
                if (was_on_disk) *was_on_disk = true;
            }
            if (!(*st = sub->impl_->import_window(k.planes, k.touched)).ok()) return nullptr;
            k.stamp = ++clock;
        }
        *st = Status::success();
        return sub;
    }

    // Survivors of the filter, as the bands' own ingests will count them (the same kernel on the same channels): what
    // points_processed grows by (src/engine/pipeline.cpp:749), whether or not any band is visited.
    Status count_kept(const PointCloud& dev, size_t* kept) {
        const size_t n = dev.count();
        *kept = n;
        if (cfg.filter.empty()) return Status::success();
        Impl::DeviceScope scope(cfg.cuda_device_id);
        std::vector<pcr_hip_predicate> preds(cfg.filter.predicates.size());
        for (size_t k = 0; k < preds.size(); ++k) {
            const FilterPredicate& pr = cfg.filter.predicates[k];
            preds[k].d_channel = static_cast<const float*>(dev.channel_data(pr.channel_name));
            preds[k].op = static_cast<int32_t>(pr.op);
            preds[k].value = pr.value;
            preds[k].set_size = static_cast<int32_t>(pr.value_set.size());
            for (size_t j = 0; j < pr.value_set.size(); ++j) preds[k].set[j] = pr.value_set[j];
        }
        detail::Buffer mb;                                   // [u64 survivor count][mask bytes]
        Status s = mb.allocate(n + 16, MemoryLocation::Device);
        if (!s.ok()) return s;
        unsigned long long h_count = 0;
        if (!(s = detail::hip_status(pcr_hip_filter_mask(preds.data(), (int)preds.size(), n, static_cast<uint8_t*>(mb.data()) + 8,
                                                         static_cast<unsigned long long*>(mb.data()), nullptr))).ok()) return s;
        if (!(s = detail::hip_status(pcr_hip_memcpy_d2h(&h_count, mb.data(), sizeof h_count, nullptr))).ok()) return s;
        if (!(s = detail::hip_status(pcr_hip_stream_synchronize(nullptr))).ok()) return s;
        *kept = (size_t)h_count;
        return Status::success();
    }

    Status ingest(const PointCloud& cloud) {
        if (cloud.count() == 0) return Status::success();
        // the checks every band's ingest would make, made once and whether or not the cloud reaches a band
        Status ok = detail::validate_cloud(cfg, cloud, PCR_HIP_MAX_FILTER_SET, PCR_HIP_MAX_FILTER_PREDICATES);
        if (!ok.ok()) return ok;
        // one device copy of a host cloud for all bands (each band's kernels read every point and keep its own)
        std::unique_ptr<PointCloud> staged;
        const PointCloud* src = &cloud;
        if (cloud.location() != MemoryLocation::Device) {
            staged = cloud.to(MemoryLocation::Device);
            if (staged) src = staged.get();
        }
        // points_processed counts the filter's survivors (round 5's fuzz: it counted the cloud); none: nothing to do, and the
        // ingest is not a collection either (src/engine/pipeline.cpp:349-353)
        size_t kept = cloud.count();
        bool kept_known = cfg.filter.empty();
        if (!kept_known && src->location() == MemoryLocation::Device) {
            if (!(ok = count_kept(*src, &kept)).ok()) return ok;
            kept_known = true;
            if (kept == 0) return Status::success();
        }
        // Which bands does this cloud reach at all?  One routing pass over x, y (pcr_hip_route_count with the bands as the parts:
        // 16 B per point, ~0.2 ms per 100 M points) against ~50 ms per band VISITED (a sub-pipeline, the band's state over
        // PCIe both ways: profiles/r05_out_of_core_cost.md) -- a survey tile that covers a corner of the grid then costs its
        // own bands only.  (The filter is not applied: a superset of the kept points, so no band is skipped wrongly.)
        std::vector<uint64_t> reach(bands.size(), 1);
        size_t valid_total = 0;
        if (bands.size() <= PCR_HIP_MAX_ROUTE_PARTS && src->location() == MemoryLocation::Device) {
            Impl::DeviceScope dev(cfg.cuda_device_id);
            const GridConfig& g = cfg.grid;
            pcr_hip_grid hg{};
            hg.min_x = g.bounds.min_x; hg.min_y = g.bounds.min_y; hg.max_x = g.bounds.max_x; hg.max_y = g.bounds.max_y;
            hg.cell_size_x = g.cell_size_x; hg.cell_size_y = g.cell_size_y;
            hg.width = g.width; hg.height = g.height; hg.tile_width = g.tile_width; hg.tile_height = g.tile_height;
            hg.own_row0 = 0; hg.own_row1 = g.height; hg.state_row0 = 0; hg.state_rows = g.height;
            std::vector<int32_t> splits;
            for (const auto& bd : bands) splits.push_back(bd.first);
            splits.push_back(bands.back().second);
            detail::Buffer dest, counts;
            std::vector<uint64_t> host(bands.size(), 0);
            if (dest.allocate(src->count(), MemoryLocation::Device).ok() && counts.allocate(bands.size() * 8, MemoryLocation::Device).ok() &&
                pcr_hip_route_count(&hg, splits.data(), (int)bands.size(), src->x(), src->y(), nullptr, src->count(),
                                    static_cast<uint8_t*>(dest.data()), static_cast<unsigned long long*>(counts.data()), nullptr) == PCR_HIP_OK &&
                pcr_hip_memcpy_d2h(host.data(), counts.data(), host.size() * 8, nullptr) == PCR_HIP_OK &&
                pcr_hip_stream_synchronize(nullptr) == PCR_HIP_OK)
                reach = host;
        }
        for (size_t b = 0; b < bands.size(); ++b) {
            if (reach[b] == 0) continue;                                    // none of this cloud's points has its centre row here
            Status s = Status::success();
            bool from_disk = false;
            std::unique_ptr<Pipeline> sub = visit(b, &s, &from_disk);
            if (!sub) return s;
            if (!(s = sub->ingest(*src)).ok()) return s;
            if (!kept_known) { kept = (size_t)sub->stats().points_processed; kept_known = true; }   // (a fresh sub-pipeline: this ingest's survivors)
            const ScatterInfo here = sub->last_scatter();
            valid_total += here.points_valid;
            last = here;
            if (!parked[b].any && here.points_valid == 0) continue;        // nothing of this cloud (or any before) fell here
            Parked& k = parked[b];
            if (here.points_valid == 0 && k.any) {                           // unchanged: the parked copy (and its files) stay current
                if (from_disk) drop_host_copy(b);
                else if (!(s = evict()).ok()) return s;
                continue;
            }
            if (!(s = sub->impl_->export_window(k.planes, k.touched)).ok()) return s;
            if (!(s = account(b)).ok()) return s;
        }
        last.points_in = cloud.count();                                     // (of the whole ingest: every band saw every point)
        last.points_valid = valid_total;
        ++collections;
        points += kept;
        if (callback) {
            ProgressInfo info = stats();
            if (!callback(info)) return Status::error(StatusCode::InvalidArgument, "pipeline: cancelled by user");
        }
        return Status::success();
    }

    Status finalize() {
        const GridConfig& g = cfg.grid;
        std::vector<BandDesc> descs;
        for (const auto& r : cfg.reductions) {
            BandDesc d;
            d.name = detail::default_band_name(r);
            d.dtype = DataType::Float32;
            d.is_state = false;
            descs.push_back(d);
        }
        if (descs.empty()) return Status::error(StatusCode::OutOfMemory, "pipeline: failed to allocate result grid");
        if (!result) result = Grid::create(g.width, g.height, descs, MemoryLocation::Host);
        if (!result) return Status::error(StatusCode::OutOfMemory, "pipeline: failed to allocate result grid");
        tiles_active = 0;
        for (size_t b = 0; b < bands.size(); ++b) {
            Status s = Status::success();
            if (!parked[b].any) {
                // no point ever had its centre row here: every tile of the band is untouched, every band NaN (Q3) -- no visit
                const size_t first = (size_t)bands[b].first * g.width, cells = (size_t)(bands[b].second - bands[b].first) * g.width;
                for (size_t o = 0; o < descs.size(); ++o)
                    std::fill_n(result->band_f32((int)o) + first, cells, std::numeric_limits<float>::quiet_NaN());
                continue;
            }
            bool from_disk = false;
            std::unique_ptr<Pipeline> sub = visit(b, &s, &from_disk);
            if (!sub) return s;
            if (!(s = sub->finalize()).ok()) return s;
            const Grid* part = sub->result();
            const int rows = bands[b].second - bands[b].first;
            if (!part || part->rows() != rows) return Status::error(StatusCode::CudaError, "pipeline: out-of-core band returned no result");
            for (size_t o = 0; o < descs.size(); ++o)
                std::copy_n(part->band_f32((int)o), (size_t)rows * g.width, result->band_f32((int)o) + (size_t)bands[b].first * g.width);
            for (uint32_t t : parked[b].touched) tiles_active += t ? 1 : 0;
            // finalize changes no state: a band that was read back from its files goes back to being "on disk" at once, and
            // the host budget holds during finalize as it does during ingest (it used to end with every band in RAM)
            if (from_disk) drop_host_copy(b);
            else if (!(s = evict()).ok()) return s;
        }
        finalized = true;
        if (!cfg.output_path.empty()) return write_geotiff(cfg.output_path, *result, cfg.grid, GeoTiffOptions());
        return Status::success();
    }

    // ---- `.pcrt` checkpoints of the whole pipeline: every band's tiles under `dir` (the in-core pipeline's layout)
    Status save_state(const std::string& dir_in) {
        const std::string dir = dir_in.empty() ? cfg.state_dir : dir_in;
        if (dir.empty()) return Status::error(StatusCode::InvalidArgument, "pipeline: no state directory given");
        for (size_t b = 0; b < bands.size(); ++b) {
            Parked& k = parked[b];
            if (!k.any) continue;
            bool from_disk = false;
            Status s = Status::success();
            if (k.on_disk) {
                if (!(s = reload(b)).ok()) return s;
                from_disk = true;
            }
            if (!(s = detail::write_state_tiles(cfg.grid, grouping.outputs, window_of(b, k), k.touched, dir)).ok()) return s;
            if (from_disk) drop_host_copy(b);
        }
        return Status::success();
    }

    Status load_state(const std::string& dir_in) {
        const std::string dir = dir_in.empty() ? cfg.state_dir : dir_in;
        if (dir.empty()) return Status::error(StatusCode::InvalidArgument, "pipeline: no state directory given");
        for (size_t b = 0; b < bands.size(); ++b) {
            Parked& k = parked[b];
            Status s = Status::success();
            Parked fresh;
            Parked* into = &k;
            if (k.any) {
                if (k.on_disk && !(s = reload(b)).ok()) return s;           // files overlay the band's current state, as in core
            } else {
                blank(b, fresh);
                into = &fresh;
            }
            size_t loaded = 0;
            if (!(s = detail::read_state_tiles(cfg.grid, grouping.outputs, window_of(b, *into), into->touched, dir, &loaded)).ok()) return s;
            if (!loaded) continue;
            if (into == &fresh) { k.planes = std::move(fresh.planes); k.touched = std::move(fresh.touched); }
            if (!(s = account(b)).ok()) return s;
        }
        return Status::success();
    }

    ProgressInfo stats() const {
        ProgressInfo info;
        info.collections_processed = collections;
        info.points_processed = points;
        info.tiles_active = tiles_active;
        info.elapsed_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
        return info;
    }
};

Pipeline::~Pipeline() = default;

bool Pipeline::out_of_core() const { return banded_ != nullptr; }

std::unique_ptr<Pipeline> Pipeline::create(const PipelineConfig& config) {
    auto p = std::unique_ptr<Pipeline>(new Pipeline());
    auto fail_with = [](const Status& s) {
        g_create_error = s.message;
        std::fprintf(stderr, "Error: %s\n", s.message.c_str());    // loud: there is no fallback path
        return std::unique_ptr<Pipeline>();
    };
    // Out of core?  Only whole-grid pipelines (a row-block shard is somebody's band already), only when the state of the
    // whole grid exceeds the device budget.
    const GridConfig& g = config.grid;
    if (config.shard_row_begin < 0 && config.shard_row_end < 0 && config.exec_mode != ExecutionMode::CPU && g.width > 0 && g.height > 0 &&
        g.tile_height > 0 && !config.reductions.empty() && cuda_device_available() && config.cuda_device_id >= 0 &&
        config.cuda_device_id < cuda_device_count()) {
        size_t budget = config.gpu_memory_budget;
        if (budget == 0) {
            size_t free_mem = 0, total_mem = 0;
            if (cuda_get_memory_info(&free_mem, &total_mem, config.cuda_device_id)) budget = (size_t)(free_mem * 0.8);
        }
        const size_t per_row = Banded::bytes_per_row(config);
        if (budget > 0 && per_row * (size_t)g.height > budget) {
            for (const auto& r : config.reductions)
                if (!registered(r.type)) return fail_with(Status::error(StatusCode::InvalidArgument, "pipeline: unknown reduction type"));
            const int th = g.tile_height;
            const size_t fit = budget / std::max<size_t>(per_row * (size_t)th, 1);          // whole tile rows that fit
            const int band_rows = (int)std::min<size_t>(std::max<size_t>(fit, 1) * (size_t)th, (size_t)g.height + th);
            if (config.result_location == MemoryLocation::Device)
                return fail_with(Status::error(StatusCode::InvalidArgument,
                    "pipeline: the grid's state (" + std::to_string(per_row * (size_t)g.height >> 20) + " MB) exceeds the device budget (" +
                    std::to_string(budget >> 20) + " MB): it is processed out of core, which needs result_location = Host"));
            p->banded_ = std::make_unique<Banded>();
            Banded& bd = *p->banded_;
            bd.cfg = config;
            bd.grouping = detail::group_reductions(config.reductions);
            for (int r0 = 0; r0 < g.height; r0 += band_rows) bd.bands.push_back({r0, std::min(r0 + band_rows, g.height)});
            bd.parked.resize(bd.bands.size());
            bd.host_budget = config.host_cache_budget;
            if (bd.host_budget == 0) {
                long pages = sysconf(_SC_AVPHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
                bd.host_budget = pages > 0 && psz > 0 ? (size_t)pages * (size_t)psz / 2 : (size_t)8 << 30;
            }
            {
                std::error_code ec;
                const std::filesystem::path base = config.state_dir.empty() ? std::filesystem::temp_directory_path(ec)
                                                                            : std::filesystem::path(config.state_dir);
                bd.spill_dir = (base / ("pcr_spill_" + std::to_string((long long)getpid()) + "_" +
                                        std::to_string((unsigned long long)(uintptr_t)p.get()))).string();
                std::filesystem::create_directories(bd.spill_dir, ec);
                if (ec) return fail_with(Status::error(StatusCode::IoError, "pipeline: cannot create " + bd.spill_dir));
            }
            // the first band is created once here, so that an impossible configuration fails at create like an in-core one
            Status s = Status::success();
            std::unique_ptr<Pipeline> probe = bd.visit(0, &s);
            if (!probe) return fail_with(s);
            std::fprintf(stderr, "Info: grid state %zu MB exceeds the device budget %zu MB - out of core in %zu bands of %d rows\n",
                         per_row * (size_t)g.height >> 20, budget >> 20, bd.bands.size(), band_rows);
            probe.reset();
            if (config.resume && !config.state_dir.empty()) {
                // the tiles under state_dir are taken band by band (and, beyond the host budget, go back there: same files)
                s = bd.load_state(config.state_dir);
                if (!s.ok()) return fail_with(s);
            }
            g_create_error.clear();
            return p;
        }
    }
    auto on_host = [&]() -> std::unique_ptr<Pipeline> {
        p->impl_.reset();
        p->host_ = std::make_unique<Host>();
        p->host_->cfg = config;
        Status hs = p->host_->init();
        if (hs.ok() && config.resume && !config.state_dir.empty()) hs = p->host_->load_state(config.state_dir);
        if (!hs.ok()) return fail_with(hs);
        g_create_error.clear();
        return std::move(p);
    };
    if (config.exec_mode == ExecutionMode::CPU) return on_host();       // "CPU mode should always work" (tests/cpp/test_error_handling.cpp:181-199)
    p->impl_ = std::make_unique<Impl>();
    p->impl_->cfg = config;
    Status s = p->impl_->init();
    if (!s.ok() && p->impl_->continue_on_host) {
        // the reference's Warning / Info line is on stderr already (Impl::init)
        const char* forbid = std::getenv("PCR_REQUIRE_GPU_ENGINE");
        if (forbid && forbid[0] && forbid[0] != '0') return fail_with(s);
        return on_host();
    }
    if (s.ok() && config.resume && !config.state_dir.empty()) s = p->impl_->load_state(config.state_dir);
    if (!s.ok()) return fail_with(s);
    g_create_error.clear();
    return p;
}

const char* Pipeline::engine() const { return host_ ? "host" : "hip"; }
int Pipeline::host_threads() const { return host_ && host_->engine ? host_->engine->threads() : 0; }
std::string Pipeline::spill_dir() const { return banded_ ? banded_->spill_dir : std::string(); }

Status Pipeline::validate() const {
    const PipelineConfig& c = host_ ? host_->cfg : banded_ ? banded_->cfg : impl_->cfg;
    if (c.grid.width <= 0 || c.grid.height <= 0)
        return Status::error(StatusCode::InvalidArgument, "pipeline: grid dimensions must be positive");
    if (c.grid.tile_width <= 0 || c.grid.tile_height <= 0)
        return Status::error(StatusCode::InvalidArgument, "pipeline: tile dimensions must be positive");
    if (c.reductions.empty())
        return Status::error(StatusCode::InvalidArgument, "pipeline: at least one reduction must be specified");
    for (const auto& r : c.reductions) {
        if (r.value_channel.empty())
            return Status::error(StatusCode::InvalidArgument, "pipeline: value_channel must be specified");
        if (!registered(r.type))
            return Status::error(StatusCode::InvalidArgument, "pipeline: unknown reduction type");
    }
    return Status::success();
}

Status Pipeline::ingest(const PointCloud& cloud) { return host_ ? host_->ingest(cloud) : banded_ ? banded_->ingest(cloud) : impl_->ingest(cloud); }
Status Pipeline::ingest_async(const PointCloud& cloud) {
    return host_ ? host_->ingest(cloud) : banded_ ? banded_->ingest(cloud) : impl_->ingest(cloud, false);
}

Status Pipeline::ingest_file(const std::string& path, size_t chunk_points, size_t* points_read) {
    if (points_read) *points_read = 0;
    if (chunk_points == 0) return Status::error(StatusCode::InvalidArgument, "pipeline: chunk_points must be positive");
    auto reader = PointCloudReader::open(path);
    if (!reader) return Status::error(StatusCode::IoError, "pipeline: failed to open point cloud file: " + path);
    chunk_points = std::min(chunk_points, std::max<size_t>(reader->info().num_points, 1));
    const MemoryLocation where = host_ ? MemoryLocation::Host : MemoryLocation::HostPinned;    // (no device: no page-locking either)
    std::unique_ptr<PointCloud> buf[2] = {PointCloud::create(chunk_points, where), PointCloud::create(chunk_points, where)};
    if (!buf[0] || !buf[1]) return Status::error(StatusCode::OutOfMemory, "pipeline: failed to allocate page-locked chunk buffers");
    size_t total = 0;
    int cur = 0;
    size_t got = reader->read_chunk(*buf[cur], chunk_points);
    while (got > 0) {
        Status s = ingest_async(*buf[cur]);              // H2D + kernels of this chunk, enqueued
        if (!s.ok()) { (void)synchronize(); return s; }
        total += got;
        got = reader->read_chunk(*buf[cur ^ 1], chunk_points);   // overlaps with them
        if (!(s = synchronize()).ok()) return s;         // buf[cur] is free again
        cur ^= 1;
    }
    if (points_read) *points_read = total;
    return Status::success();
}
Status Pipeline::finalize() { return host_ ? host_->finalize() : banded_ ? banded_->finalize() : impl_->finalize(); }
Status Pipeline::finalize_async() { return host_ ? host_->finalize() : banded_ ? banded_->finalize() : impl_->finalize(false); }

Status Pipeline::run(const std::vector<const PointCloud*>& clouds) {
    for (const PointCloud* c : clouds) {
        if (!c) return Status::error(StatusCode::InvalidArgument, "pipeline: null cloud pointer");
        Status s = ingest(*c);
        if (!s.ok()) return s;
    }
    return finalize();
}

void Pipeline::set_progress_callback(ProgressCallback cb) {
    if (host_) host_->callback = std::move(cb);
    else if (banded_) banded_->callback = std::move(cb);
    else impl_->callback = std::move(cb);
}
const Grid* Pipeline::result() const {
    if (host_) return host_->finalized ? host_->result.get() : nullptr;
    if (banded_) return banded_->finalized ? banded_->result.get() : nullptr;
    return impl_->finalized ? impl_->result.get() : nullptr;
}
ProgressInfo Pipeline::stats() const { return host_ ? host_->stats() : banded_ ? banded_->stats() : impl_->stats(); }

// (out of core: no plane lives in HBM between two calls -- the shard accessors answer for "no shard")
int Pipeline::halo_rows() const { return banded_ || host_ ? 0 : impl_->halo; }
Status Pipeline::line_reach_rows(const PointCloud& cloud, int* rows) {
    if (banded_ || host_) { if (rows) *rows = 0; return Status::success(); }
    return impl_->query_line_reach(cloud, rows);
}
int Pipeline::state_row_begin() const { return banded_ || host_ ? 0 : impl_->hg.state_row0; }
int Pipeline::state_row_count() const { return banded_ || host_ ? 0 : impl_->hg.state_rows; }

std::vector<Pipeline::PlaneView> Pipeline::state_planes() const {
    std::vector<PlaneView> out;
    if (banded_ || host_) return out;
    {
        Impl::DeviceScope dev(impl_->cfg.cuda_device_id);
        bool filled = false;
        for (const auto& g : impl_->groups) filled = filled || !g.defined;
        (void)impl_->define_all_planes();               // the caller reads (and may write) them: identity where nothing was ingested
        // The fills run on the pipeline's stream and the caller may read on any other (torch's current stream, a host copy):
        // they are complete when the pointers leave.  Planes that were already defined cost no synchronisation.
        if (filled) (void)pcr_hip_stream_synchronize(impl_->stream);
    }
    for (auto& g : impl_->groups) g.fresh = false;      // mutable pointers leave the pipeline: assume the planes get written
    impl_->state_shared = true;
    impl_->bands_stale();
    for (size_t g = 0; g < impl_->groups.size(); ++g)
        for (int p = 0; p < 4; ++p)
            if (impl_->groups[g].mask & kPlaneBits[p])
                out.push_back({impl_->groups[g].planes[p].data(), (int)kPlaneBits[p], (int)g,
                               impl_->groups[g].glyph.type == GlyphType::Point ? 0 : impl_->halo});
    return out;
}

std::vector<int> Pipeline::reduction_groups() const {
    std::vector<int> out;
    if (banded_ || host_) return out;
    for (const auto& o : impl_->outputs) out.push_back(o.group);
    return out;
}

void* Pipeline::tile_touched_device(int* tiles_x, int* tiles_y) const {
    if (banded_ || host_) return nullptr;
    uint32_t* d = nullptr;
    int32_t tx = 0, ty = 0;
    if (pcr_hip_engine_tile_touched(impl_->engine, &d, &tx, &ty) != PCR_HIP_OK) return nullptr;
    impl_->state_shared = true;                         // (the flags may be written from outside: the shard exchange does)
    impl_->bands_stale();
    if (tiles_x) *tiles_x = tx;
    if (tiles_y) *tiles_y = ty;
    return d;
}

const void* Pipeline::tile_touched_device_readonly(int* tiles_x, int* tiles_y) const {
    if (banded_ || host_) return nullptr;
    uint32_t* d = nullptr;
    int32_t tx = 0, ty = 0;
    if (pcr_hip_engine_tile_touched(impl_->engine, &d, &tx, &ty) != PCR_HIP_OK) return nullptr;
    if (tiles_x) *tiles_x = tx;
    if (tiles_y) *tiles_y = ty;
    return d;
}

Status Pipeline::merge_touched(const void* d_union) {
    if (banded_ || host_) return Status::error(StatusCode::NotImplemented, "pipeline: an out-of-core or host-engine pipeline is not a shard");
    if (!d_union) return Status::error(StatusCode::InvalidArgument, "pipeline: merge_touched: null flags");
    Impl::DeviceScope dev(impl_->cfg.cuda_device_id);
    uint32_t* d = nullptr;
    int32_t tx = 0, ty = 0;
    Status s = detail::hip_status(pcr_hip_engine_tile_touched(impl_->engine, &d, &tx, &ty));
    if (!s.ok()) return s;
    // (the stored bands' device words decide: the host keeps offering them to finalize, whose kernel runs when they are 0)
    // only a flag of a tile this device owns rows of can change its bands (the union used to drop the stored
    // bands whenever ANY tile of the grid changed, i.e. always, on shards that do not share every tile)
    const int th = impl_->cfg.grid.tile_height;
    const int t0 = impl_->hg.own_row0 / th, t1 = std::min(ty, (impl_->hg.own_row1 + th - 1) / th);
    return detail::hip_status(pcr_hip_touched_union_owned(d, static_cast<const uint32_t*>(d_union), tx, ty, std::min(t0, t1), t1,
                                                          static_cast<uint32_t*>(impl_->d_bands_done.data()),
                                                          (int32_t)impl_->groups.size(), impl_->stream));
}

Status Pipeline::save_state(const std::string& dir) {
    if (host_) return host_->save_state(dir);
    if (banded_) return banded_->save_state(dir);
    return impl_->save_state(dir);
}
Status Pipeline::load_state(const std::string& dir) {
    if (host_) return host_->load_state(dir);
    if (banded_) return banded_->load_state(dir);
    return impl_->load_state(dir);
}

const float* Pipeline::result_band_device(int band) const {
    if (banded_ || host_ || !impl_->finalized || !impl_->result || band < 0 || band >= (int)impl_->outputs.size()) return nullptr;
    return impl_->band_device((size_t)band);
}

Status Pipeline::synchronize() { return banded_ || host_ ? Status::success() : detail::hip_status(pcr_hip_stream_synchronize(impl_->stream)); }
void* Pipeline::stream_handle() const { return banded_ || host_ ? nullptr : impl_->stream; }

void Pipeline::profile_enable(bool on, const std::string& only_kernel) {
    if (banded_ || host_) return;
    pcr_hip_engine_profile_only(impl_->engine, only_kernel.c_str());
    pcr_hip_engine_profile_enable(impl_->engine, on ? 1 : 0);
}

std::vector<Pipeline::KernelTime> Pipeline::profile_read(bool reset) {
    std::vector<KernelTime> out;
    if (banded_ || host_) return out;
    pcr_hip_kernel_time buf[32];
    int n = 0;
    if (pcr_hip_engine_profile_read(impl_->engine, buf, 32, &n, reset ? 1 : 0) != PCR_HIP_OK) return out;
    for (int i = 0; i < std::min(n, 32); ++i) out.push_back({buf[i].name, buf[i].launches, buf[i].total_ms});
    return out;
}

Pipeline::ScatterInfo Pipeline::last_scatter() const {
    if (host_) return host_->last;
    if (banded_) return banded_->last;
    pcr_hip_scatter_stats st{};
    pcr_hip_engine_stats(impl_->engine, &st);
    int fused = 0;
    for (const auto& g : impl_->groups) fused += g.bands_with_scatter ? 1 : 0;
    return {st.path, st.lds_tile_w, st.lds_tile_h, st.lds_apron, st.num_bins,
            (size_t)st.points_in, (size_t)st.points_valid, st.scatter_chunk, fused};
}

}  // namespace pcr
