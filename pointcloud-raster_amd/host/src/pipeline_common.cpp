// pipeline_common.cpp -- what the engines behind pcr::Pipeline share: the cloud checks, `.pcrt` tile files of a window of state planes (see pipeline_common.h).
#include "pipeline_common.h"

#include "pcr/core/point_cloud.h"
#include "pcr/io/tile_state_io.h"

#include <algorithm>
#include <filesystem>

namespace pcr {
namespace detail {

Status validate_cloud(const PipelineConfig& cfg, const PointCloud& cloud, size_t max_set, size_t max_predicates) {
    for (const auto& pr : cfg.filter.predicates) {
        if (!cloud.channel_data(pr.channel_name))
            return Status::error(StatusCode::InvalidArgument, "filter_points: channel not found: " + pr.channel_name);
        const ChannelDesc* d = cloud.channel(pr.channel_name);
        if (!d || d->dtype != DataType::Float32)
            return Status::error(StatusCode::InvalidArgument, "filter_points: only Float32 channels supported for filtering");
        if (max_set && pr.value_set.size() > max_set)
            return Status::error(StatusCode::InvalidArgument,
                                 "filter_points: value_set larger than " + std::to_string(max_set) + " entries is not supported on the device");
    }
    if (max_predicates && cfg.filter.predicates.size() > max_predicates)
        return Status::error(StatusCode::InvalidArgument, "filter_points: more than " + std::to_string(max_predicates) + " predicates");
    for (const auto& r : cfg.reductions) {
        if (!cloud.channel_data(r.value_channel))
            return Status::error(StatusCode::InvalidArgument, "pipeline: value channel not found: " + r.value_channel);
        const ChannelDesc* d = cloud.channel(r.value_channel);
        if (!d || d->dtype != DataType::Float32)
            return Status::error(StatusCode::InvalidArgument, "pipeline: value channel must be Float32");
        if (r.glyph.type != GlyphType::Point && !glyph_reduction_ok(r.type))
            return Status::error(StatusCode::NotImplemented,
                "pipeline: glyph splatting only supports WeightedAverage, Average, Sum, or Count reduction types");
    }
    return Status::success();
}

std::string reduction_state_dir(const std::string& dir, size_t r, size_t n_outputs) {
    return n_outputs == 1 ? dir : dir + "/reduction_" + std::to_string(r);
}

namespace {
struct TileRect { int c0, r0, nc, nr; bool inside; };

TileRect tile_rect(const GridConfig& g, const StateWindow& w, int tx, int ty) {
    TileRect t;
    t.c0 = tx * g.tile_width;
    t.r0 = ty * g.tile_height;
    t.nc = std::min(g.tile_width, g.width - t.c0);
    t.nr = std::min(g.tile_height, g.height - t.r0);
    const int lo = w.own_row0 >= 0 ? w.own_row0 : w.row0, hi = w.own_row0 >= 0 ? w.own_row1 : w.row0 + w.rows;
    t.inside = t.r0 >= lo && t.r0 + t.nr <= hi && t.r0 >= w.row0 && t.r0 + t.nr <= w.row0 + w.rows;
    return t;
}
}  // namespace

Status write_state_tiles(const GridConfig& g, const std::vector<StateOutput>& outputs, const StateWindow& w,
                         const std::vector<uint32_t>& touched, const std::string& dir, std::vector<std::string>* written) {
    const int tiles_x = (g.width + g.tile_width - 1) / g.tile_width;
    const int tiles_y = (g.height + g.tile_height - 1) / g.tile_height;
    std::error_code ec;
    std::vector<float> buf;
    for (size_t r = 0; r < outputs.size(); ++r) {
        const std::string rdir = reduction_state_dir(dir, r, outputs.size());
        std::filesystem::create_directories(rdir, ec);
        if (ec) return Status::error(StatusCode::IoError, "pipeline: cannot create " + rdir);
        int pl[2];
        const int k = state_planes_of(outputs[r].type, pl);
        for (int ty = 0; ty < tiles_y; ++ty)
            for (int tx = 0; tx < tiles_x; ++tx) {
                if (!touched[(size_t)ty * tiles_x + tx]) continue;          // only tiles that have state
                const TileRect t = tile_rect(g, w, tx, ty);
                if (!t.inside) continue;
                buf.resize((size_t)k * t.nc * t.nr);
                for (int f = 0; f < k; ++f) {
                    const float* plane = w.plane(outputs[r].group, pl[f]);
                    if (!plane) return Status::error(StatusCode::InvalidArgument, "pipeline: a reduction's state plane is missing");
                    for (int y = 0; y < t.nr; ++y)
                        std::copy_n(plane + (size_t)(t.r0 - w.row0 + y) * g.width + t.c0, t.nc, buf.data() + ((size_t)f * t.nr + y) * t.nc);
                }
                TileIndex ti;
                ti.row = ty;
                ti.col = tx;
                const std::string path = tile_state_filename(rdir, ti);
                Status s = write_tile_state(path, ti, t.nc, t.nr, k, outputs[r].type, buf.data());
                if (!s.ok()) return s;
                if (written) written->push_back(path);
            }
    }
    return Status::success();
}

Status read_state_tiles(const GridConfig& g, const std::vector<StateOutput>& outputs, const StateWindow& w,
                        std::vector<uint32_t>& touched, const std::string& dir, size_t* loaded) {
    const int tiles_x = (g.width + g.tile_width - 1) / g.tile_width;
    const int tiles_y = (g.height + g.tile_height - 1) / g.tile_height;
    // rows this window takes from the files: its own rows (a shard never loads into its apron), else the whole window.  A
    // tile that the range only CUTS is read whole and its rows inside the range are taken: a row-block shard resumes from any
    // checkpoint of the grid, whoever wrote it.
    const int lo = w.own_row0 >= 0 ? std::max(w.own_row0, w.row0) : w.row0;
    const int hi = w.own_row0 >= 0 ? std::min(w.own_row1, w.row0 + w.rows) : w.row0 + w.rows;
    std::vector<float> buf;
    size_t taken = 0;
    for (size_t r = 0; r < outputs.size(); ++r) {
        const std::string rdir = reduction_state_dir(dir, r, outputs.size());
        int pl[2];
        const int k = state_planes_of(outputs[r].type, pl);
        for (int ty = 0; ty < tiles_y; ++ty)
            for (int tx = 0; tx < tiles_x; ++tx) {
                const TileRect t = tile_rect(g, w, tx, ty);
                const int y0 = std::max(t.r0, lo), y1 = std::min(t.r0 + t.nr, hi);
                if (y0 >= y1) continue;
                TileIndex ti;
                ti.row = ty;
                ti.col = tx;
                const std::string path = tile_state_filename(rdir, ti);
                std::error_code ec;
                if (!std::filesystem::exists(path, ec)) continue;
                TileIndex ft;
                int fc = 0, fr = 0, fk = 0;
                ReductionType ftype;
                if (!read_tile_state_header(path, ft, fc, fr, fk, ftype).ok()) continue;
                if (fc != t.nc || fr != t.nr || fk != k || ftype != outputs[r].type || ft.row != ty || ft.col != tx) continue;
                buf.resize((size_t)k * t.nc * t.nr);
                if (!read_tile_state(path, ft, fc, fr, fk, ftype, buf.data()).ok()) continue;
                for (int f = 0; f < k; ++f) {
                    float* plane = w.plane(outputs[r].group, pl[f]);
                    if (!plane) return Status::error(StatusCode::InvalidArgument, "pipeline: a reduction's state plane is missing");
                    for (int y = y0; y < y1; ++y)
                        std::copy_n(buf.data() + ((size_t)f * t.nr + (y - t.r0)) * t.nc, t.nc, plane + (size_t)(y - w.row0) * g.width + t.c0);
                }
                touched[(size_t)ty * tiles_x + tx] = 1;
                ++taken;
            }
    }
    if (loaded) *loaded = taken;
    return Status::success();
}

}  // namespace detail
}  // namespace pcr
