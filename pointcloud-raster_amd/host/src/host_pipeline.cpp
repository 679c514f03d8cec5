// host_pipeline.cpp -- see host_pipeline.h.
#include "host_pipeline.h"

#include "pcr/core/point_cloud.h"
#include "pcr/io/grid_io.h"

#include <algorithm>
#include <map>

namespace pcr {

using detail::glyph_reduction_ok;
using detail::planes_for;
using detail::registered;
using detail::same_glyph;

Status Pipeline::Host::init() {
    const GridConfig& g = cfg.grid;
    if (g.width <= 0 || g.height <= 0)
        return Status::error(StatusCode::InvalidArgument, "pipeline: grid dimensions must be positive");
    if (g.tile_width <= 0 || g.tile_height <= 0)
        return Status::error(StatusCode::InvalidArgument, "pipeline: tile dimensions must be positive");
    for (const auto& r : cfg.reductions)
        if (!registered(r.type)) return Status::error(StatusCode::InvalidArgument, "pipeline: unknown reduction type");
    if (cfg.shard_row_begin >= 0 || cfg.shard_row_end >= 0)
        return Status::error(StatusCode::InvalidArgument, "pipeline: row-block shards run on the GPU engine only (one process per GPU)");
    if (cfg.result_location == MemoryLocation::Device)
        return Status::error(StatusCode::InvalidArgument, "pipeline: the CPU engine's result lives in host memory (result_location = Device asked)");
    engine = std::make_unique<detail::HostEngine>(g, (int)cfg.cpu_threads);        // cpu_threads = 0: every core (pipeline.h:78)
    for (const auto& r : cfg.reductions) {
        int gi = -1;
        for (size_t k = 0; k < groups.size(); ++k)
            if (groups[k].value_channel == r.value_channel && same_glyph(groups[k].glyph, r.glyph)) gi = (int)k;
        if (gi < 0) {
            groups.emplace_back();
            gi = (int)groups.size() - 1;
            groups[(size_t)gi].value_channel = r.value_channel;
            groups[(size_t)gi].glyph = r.glyph;
        }
        groups[(size_t)gi].mask |= planes_for(r.type);
        Output o;
        o.group = gi;
        o.type = r.type;
        o.band_name = detail::default_band_name(r);
        outputs.push_back(o);
    }
    // Tile state is identity-initialised (src/engine/tile_manager.cpp:183-260); here for the whole grid at once.
    for (auto& gr : groups) engine->init_planes(gr.planes, gr.mask);
    return Status::success();
}

Status Pipeline::Host::ingest(const PointCloud& cloud_in) {
    const size_t n = cloud_in.count();
    if (n == 0) return Status::success();                            // pipeline.cpp:284-287
    std::unique_ptr<PointCloud> copy;
    const PointCloud* cloud = &cloud_in;
    if (cloud->location() == MemoryLocation::Device) {
        copy = cloud->to(MemoryLocation::Host);
        if (!copy) return Status::error(StatusCode::OutOfMemory, "pipeline: cannot copy the device cloud to the host");
        cloud = copy.get();
    }
    // the reference's checks and messages (pipeline_common.cpp; no device limits on the host)
    {
        Status ok = detail::validate_cloud(cfg, *cloud, 0, 0);
        if (!ok.ok()) return ok;
    }
    auto f32 = [&](const std::string& name) -> const float* {
        if (name.empty()) return nullptr;
        const ChannelDesc* d = cloud->channel(name);
        if (!d || d->dtype != DataType::Float32) return nullptr;     // -> GlyphSpec default
        return cloud->channel_f32(name);
    };

    // Filter stage: AND of the predicates (evaluate_predicate, src/engine/filter.cpp:34-56).  As in the HIP pipeline a
    // filtered-out point does not exist for any reduction (the reference routes the unfiltered cloud: DESIGN section 1).
    std::vector<uint8_t> keep;
    size_t kept = n;
    if (!cfg.filter.empty()) {
        keep.assign(n, 1);
        for (const auto& pr : cfg.filter.predicates) {
            const float* ch = cloud->channel_f32(pr.channel_name);
            const int team = engine->threads();
#pragma omp parallel for num_threads(team) schedule(static)
            for (int64_t i = 0; i < (int64_t)n; ++i) {
                const float v = ch[i];
                bool ok = false;
                switch (pr.op) {
                    case CompareOp::Equal: ok = v == pr.value; break;
                    case CompareOp::NotEqual: ok = v != pr.value; break;
                    case CompareOp::Less: ok = v < pr.value; break;
                    case CompareOp::LessEqual: ok = v <= pr.value; break;
                    case CompareOp::Greater: ok = v > pr.value; break;
                    case CompareOp::GreaterEqual: ok = v >= pr.value; break;
                    case CompareOp::InSet:
                    case CompareOp::NotInSet: {
                        bool in = false;
                        for (float sv : pr.value_set) in = in || v == sv;
                        ok = pr.op == CompareOp::InSet ? in : !in;
                        break;
                    }
                }
                if (!ok) keep[(size_t)i] = 0;
            }
        }
        kept = 0;
        for (uint8_t k : keep) kept += k;
        if (kept == 0) return Status::success();                     // pipeline.cpp:349-353
    }

    // the engine indexes a cloud with 32 bits: larger ones go in slices
    const size_t slice = (size_t)1 << 31;
    size_t valid = 0;
    for (size_t i0 = 0; i0 < n; i0 += slice) {
        const size_t m = std::min(slice, n - i0);
        valid += engine->route(cloud->x() + i0, cloud->y() + i0, keep.empty() ? nullptr : keep.data() + i0, m);
        for (auto& gr : groups) {
            const float* v = f32(gr.value_channel);
            if (gr.glyph.type == GlyphType::Point) {
                engine->scatter_point(gr.planes, v + i0);
            } else {
                detail::HostGlyphArrays arr;
                auto at = [&](const std::string& name) { const float* p = f32(name); return p ? p + i0 : nullptr; };
                arr.direction = at(gr.glyph.direction_channel);
                arr.half_length = at(gr.glyph.half_length_channel);
                arr.sigma_x = at(gr.glyph.sigma_x_channel);
                arr.sigma_y = at(gr.glyph.sigma_y_channel);
                arr.rotation = at(gr.glyph.rotation_channel);
                detail::HostPlanes& pl = gr.planes;
                const uint32_t keep_mask = pl.mask;
                pl.mask &= 3u;                                        // glyph reductions feed the sum and weight planes only
                engine->scatter_glyph(pl, gr.glyph, arr, v + i0);
                pl.mask = keep_mask;
            }
        }
    }
    last = ScatterInfo{};
    last.path = -1;                                                  // neither of the device's scatter paths
    last.points_in = n;
    last.points_valid = valid;
    points += kept;                                                  // points_processed += filtered_count (pipeline.cpp:749)
    collections++;
    if (callback) {
        ProgressInfo info = stats();
        if (!callback(info)) return Status::error(StatusCode::InvalidArgument, "pipeline: cancelled by user");
    }
    return Status::success();
}

Status Pipeline::Host::finalize() {
    if (outputs.empty()) return Status::error(StatusCode::OutOfMemory, "pipeline: failed to allocate result grid");
    const GridConfig& g = cfg.grid;
    if (!result) {
        std::vector<BandDesc> bands;
        for (const auto& o : outputs) {                              // band naming: pipeline.cpp:1175-1186
            BandDesc b;
            b.name = o.band_name;
            b.dtype = DataType::Float32;
            b.is_state = false;
            bands.push_back(b);
        }
        result = Grid::create(g.width, g.height, bands, MemoryLocation::Host);
        if (!result) return Status::error(StatusCode::OutOfMemory, "pipeline: failed to allocate result grid");
    }
    for (size_t r = 0; r < outputs.size(); ++r)
        engine->finalize(groups[(size_t)outputs[r].group].planes, outputs[r].type, result->band_f32((int)r));
    finalized = true;
    if (!cfg.output_path.empty()) return write_geotiff(cfg.output_path, *result, g, GeoTiffOptions());    // pipeline.cpp:1351-1361
    return Status::success();
}

std::vector<detail::StateOutput> Pipeline::Host::state_outputs() const {
    std::vector<detail::StateOutput> o;
    for (const auto& out : outputs) o.push_back({out.group, out.type});
    return o;
}

Status Pipeline::Host::save_state(const std::string& dir_in) {
    const std::string dir = dir_in.empty() ? cfg.state_dir : dir_in;
    if (dir.empty()) return Status::error(StatusCode::InvalidArgument, "pipeline: no state directory given");
    detail::StateWindow w;
    w.row0 = 0;
    w.rows = cfg.grid.height;
    w.plane = [&](int g, int p) -> float* { auto& v = groups[(size_t)g].planes.plane[p]; return v.empty() ? nullptr : v.data(); };
    return detail::write_state_tiles(cfg.grid, state_outputs(), w, engine->touched(), dir);
}

Status Pipeline::Host::load_state(const std::string& dir_in) {
    const std::string dir = dir_in.empty() ? cfg.state_dir : dir_in;
    if (dir.empty()) return Status::error(StatusCode::InvalidArgument, "pipeline: no state directory given");
    detail::StateWindow w;
    w.row0 = 0;
    w.rows = cfg.grid.height;
    w.plane = [&](int g, int p) -> float* { auto& v = groups[(size_t)g].planes.plane[p]; return v.empty() ? nullptr : v.data(); };
    return detail::read_state_tiles(cfg.grid, state_outputs(), w, engine->touched(), dir, nullptr);
}

ProgressInfo Pipeline::Host::stats() const {
    ProgressInfo info;
    info.collections_processed = collections;
    info.collections_total = 0;
    info.points_processed = points;
    info.tiles_active = engine ? engine->tiles_active() : 0;
    info.elapsed_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
    return info;
}

}  // namespace pcr
