// sharded_pipeline.cpp -- see pcr/engine/sharded_pipeline.h.
#include "pcr/engine/sharded_pipeline.h"

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "buffer.h"
#include "pcr/core/grid.h"
#include "pcr/core/point_cloud.h"
#include "pcr/io/grid_io.h"
#include "pcr_hip.h"
#include "pipeline_common.h"

namespace pcr {

namespace {
thread_local std::string g_create_error;
}

const std::string& ShardedPipeline::create_error() { return g_create_error; }

Status ShardedPipeline::make_id(uint8_t* id128) { return detail::hip_status(pcr_hip_comm_unique_id(id128)); }

std::pair<int, int> ShardedPipeline::row_block(int rank, int world, int height, int align) {
    const int units = (height + align - 1) / align;
    const int base = units / world, extra = units % world;
    const int u0 = rank * base + std::min(rank, extra);
    const int u1 = u0 + base + (rank < extra ? 1 : 0);
    return {std::min(u0 * align, height), std::min(u1 * align, height)};
}

std::unique_ptr<ShardedPipeline> ShardedPipeline::create(PipelineConfig cfg, const uint8_t* id128, int rank, int world,
                                                         int device, int align) {
    g_create_error.clear();
    if (world < 1 || rank < 0 || rank >= world || !id128) {
        g_create_error = "ShardedPipeline: rank outside [0, world) or null id";
        return nullptr;
    }
    std::unique_ptr<ShardedPipeline> sp(new ShardedPipeline());
    sp->rank_ = rank;
    sp->world_ = world;
    const auto blk = row_block(rank, world, cfg.grid.height, align);
    sp->r0_ = blk.first;
    sp->r1_ = blk.second;
    cfg.shard_row_begin = blk.first;
    cfg.shard_row_end = blk.second;
    cfg.cuda_device_id = device;
    sp->grid_ = cfg.grid;
    sp->height_ = cfg.grid.height;
    sp->align_ = align;
    // ONE file for the whole grid, written by rank 0 from the gathered strips (the reference writes one file,
    // src/engine/pipeline.cpp:1351-1361) -- not a strip per rank under the same name
    sp->output_path_ = cfg.output_path;
    cfg.output_path.clear();
    sp->state_dir_ = cfg.state_dir;
    sp->reductions_ = cfg.reductions;
    sp->pipe_ = Pipeline::create(cfg);
    if (!sp->pipe_) {
        g_create_error = "ShardedPipeline: " + pipeline_create_error();
        return nullptr;
    }
    sp->halo_ = sp->pipe_->halo_rows();
    sp->width_ = cfg.grid.width;
    const int th = cfg.grid.tile_height;
    sp->tiles_local_ = true;
    for (int r = 0; r < world; ++r) {
        const int b0 = row_block(r, world, cfg.grid.height, align).first;
        if (b0 % th != 0 && b0 < cfg.grid.height) sp->tiles_local_ = false;
    }
    // Feasibility is decided from EVERY rank's block, so that create fails on every rank or on none (the same
    // configuration gives the same blocks everywhere): a rank that owns no rows, or a block shorter than the halo its
    // neighbours keep, cannot take part in the neighbour exchange (pcr_hip_comm_halo_plan would refuse it at the first
    // exchange -- on every rank as well; here it is refused before anything is ingested).
    if (world > 1 && !sp->tiles_local_) {
        for (int r = 0; r < world; ++r) {
            const auto b = row_block(r, world, cfg.grid.height, align);
            const int rows = b.second - b.first;
            if (rows <= 0 || rows < sp->halo_) {
                g_create_error = "ShardedPipeline: rank " + std::to_string(r) + " would own " + std::to_string(rows) +
                                 " rows with a halo of " + std::to_string(sp->halo_) +
                                 " rows: use fewer ranks, a smaller radius or tile-aligned blocks (refused on every rank)";
                return nullptr;
            }
        }
    }
    for (const auto& r : cfg.reductions)
        if (r.glyph.type == GlyphType::Line && !r.glyph.half_length_channel.empty()) sp->line_hl_groups_ = true;
    if (world > 1) {
        Status s = detail::hip_status(pcr_hip_comm_create(&sp->comm_, id128, rank, world, device));
        if (!s.ok()) {
            g_create_error = "ShardedPipeline: " + s.message;
            return nullptr;
        }
    }
    return sp;
}

ShardedPipeline::~ShardedPipeline() {
    if (comm_) pcr_hip_comm_destroy(comm_);
}

Status ShardedPipeline::ingest(const PointCloud& cloud) {
    // A per-point half_length Line group can need more halo rows than the shard keeps.  Pipeline::ingest would refuse
    // such a cloud -- on the ranks that hold the long segments only, while the others carry on into the exchange and wait
    // there.  So every rank's need is reduced (MAX) first and all ranks refuse the round together.
    if (world_ > 1 && comm_ && line_hl_groups_ && !tiles_local_) {
        int rows = 0;
        Status s = pipe_->line_reach_rows(cloud, &rows);
        // (a rank whose query failed still takes part in the agreement, with a need no halo can meet)
        int32_t need = s.ok() ? (int32_t)rows : INT32_MAX;
        Status a = detail::hip_status(pcr_hip_comm_agree_max_i32(comm_, &need, pipe_->stream_handle()));
        if (!a.ok()) return a;
        if (!s.ok()) return s;
        if (need > halo_)
            return Status::error(StatusCode::InvalidArgument,
                "pipeline: a Line segment of this round reaches " + std::to_string((long long)need) +
                " rows beyond its centre row on some rank, but the row-block shards keep a halo of " + std::to_string(halo_) +
                " rows; set PipelineConfig.shard_halo_rows >= " + std::to_string((long long)need) +
                " on every rank, or use tile-aligned row blocks (refused on every rank, nothing was accumulated)");
    }
    return pipe_->ingest(cloud);
}

Status ShardedPipeline::exchange() {
    // footprints are clipped to the reference tile of their centre cell (Q4): with tile-aligned blocks nothing lands
    // in a neighbour's rows and the touched flags are local
    if (world_ == 1 || tiles_local_ || !comm_) return Status::success();
    void* stream = pipe_->stream_handle();
    std::vector<pcr_hip_halo_plane> planes;
    for (const auto& v : pipe_->state_planes())
        if (v.reach_rows > 0) planes.push_back({static_cast<float*>(v.device_ptr), (uint32_t)v.plane_kind, 0u});
    if (halo_ > 0 && !planes.empty()) {
        Status s = detail::hip_status(pcr_hip_comm_halo_reduce(comm_, planes.data(), (int)planes.size(), width_,
                                                               pipe_->state_row_begin(), pipe_->state_row_count(), r0_, r1_,
                                                               halo_, stream));
        if (!s.ok()) return s;
    }
    int tx = 0, ty = 0;
    void* touched = pipe_->tile_touched_device(&tx, &ty);
    return detail::hip_status(pcr_hip_comm_allreduce_max_u32(comm_, static_cast<uint32_t*>(touched), tx * ty, stream));
}

Status ShardedPipeline::finalize() {
    Status s = exchange();
    if (!s.ok()) return s;
    if (!(s = pipe_->finalize()).ok()) return s;
    if (output_path_.empty()) return s;
    std::unique_ptr<Grid> whole;
    if (!(s = gather(0, &whole)).ok()) return s;
    if (rank_ != 0) return s;
    return write_geotiff(output_path_, *whole, grid_, GeoTiffOptions());
}

Status ShardedPipeline::save_state(const std::string& dir_in) {
    Status s = exchange();                    // what the apron rows hold belongs in the neighbour's tiles
    if (!s.ok()) return s;
    if (world_ == 1 || tiles_local_) return pipe_->save_state(dir_in);      // every tile has one owner: each rank writes its own
    // Blocks that cut reference tiles: a tile has two owners and a `.pcrt` file holds a whole tile, so the owned rows of every
    // plane travel to rank 0 (pcr_hip_comm_gatherv: the strips in rank order are the grid's rows in order), which writes the
    // checkpoint of the whole grid -- the files an unsharded pipeline would have written.
    const std::string dir = dir_in.empty() ? state_dir_ : dir_in;
    if (dir.empty()) return Status::error(StatusCode::InvalidArgument, "pipeline: no state directory given");
    void* stream = pipe_->stream_handle();
    const std::vector<Pipeline::PlaneView> views = pipe_->state_planes();
    const detail::Grouping grouping = detail::group_reductions(reductions_);
    const uint64_t strip = (uint64_t)(r1_ - r0_) * (uint64_t)width_, whole = (uint64_t)height_ * (uint64_t)width_;
    const bool root = rank_ == 0;
    std::vector<std::vector<float>> planes(grouping.masks.size() * 4);
    detail::Buffer landing;
    Status local = root ? landing.allocate((size_t)whole * sizeof(float), MemoryLocation::Device) : Status::success();
    for (const auto& v : views) {
        int p = 0;
        while (p < 4 && detail::kPlaneBits[p] != (uint32_t)v.plane_kind) ++p;
        if (p == 4 || v.group < 0 || (size_t)v.group >= grouping.masks.size())
            return Status::error(StatusCode::InvalidArgument, "ShardedPipeline::save_state: unknown plane");
        const void* send[1] = {static_cast<const float*>(v.device_ptr) + (size_t)(r0_ - pipe_->state_row_begin()) * (size_t)width_};
        void* recv[1] = {root && local.ok() ? landing.data() : nullptr};
        const int32_t elem[1] = {4};
        uint64_t counts[PCR_HIP_MAX_ROUTE_PARTS] = {};
        s = detail::hip_status(pcr_hip_comm_gatherv(comm_, 1, send, recv, elem, strip, root ? whole : 0, counts, 0, stream));
        if (!s.ok()) return local.ok() ? s : local;
        if (!root) continue;
        std::vector<float>& host = planes[(size_t)v.group * 4 + (size_t)p];
        host.resize((size_t)whole);
        if (!(s = detail::hip_status(pcr_hip_memcpy_d2h(host.data(), landing.data(), (size_t)whole * sizeof(float), stream))).ok()) return s;
        if (!(s = detail::hip_status(pcr_hip_stream_synchronize(stream))).ok()) return s;     // (the landing area is reused)
    }
    if (!root) return Status::success();
    int tx = 0, ty = 0;
    const void* d_touched = pipe_->tile_touched_device_readonly(&tx, &ty);                    // the union, after the exchange
    std::vector<uint32_t> touched((size_t)tx * (size_t)ty);
    if (!d_touched) return Status::error(StatusCode::CudaError, "ShardedPipeline::save_state: no touched-tile flags");
    if (!(s = detail::hip_status(pcr_hip_memcpy_d2h(touched.data(), d_touched, touched.size() * 4, stream))).ok()) return s;
    if (!(s = detail::hip_status(pcr_hip_stream_synchronize(stream))).ok()) return s;
    detail::StateWindow w;
    w.row0 = 0;
    w.rows = height_;
    w.plane = [&planes](int g, int p) -> float* {
        auto& v = planes[(size_t)g * 4 + (size_t)p];
        return v.empty() ? nullptr : v.data();
    };
    return detail::write_state_tiles(grid_, grouping.outputs, w, touched, dir);
}

Status ShardedPipeline::gather(int dst_rank, std::unique_ptr<Grid>* out) {
    if (!out) return Status::error(StatusCode::InvalidArgument, "ShardedPipeline::gather: null result pointer");
    out->reset();
    if (dst_rank < 0 || dst_rank >= world_)
        return Status::error(StatusCode::InvalidArgument, "ShardedPipeline::gather: destination rank outside [0, world)");
    const Grid* part = pipe_->result();
    if (!part) return Status::error(StatusCode::InvalidArgument, "ShardedPipeline::gather: finalize() first");
    const int nb = part->num_bands();
    if (world_ == 1) {
        *out = part->to(MemoryLocation::Host);
        return *out ? Status::success() : Status::error(StatusCode::OutOfMemory, "ShardedPipeline::gather: host allocation failed");
    }
    void* stream = pipe_->stream_handle();
    const uint64_t strip = (uint64_t)(r1_ - r0_) * (uint64_t)width_, whole = (uint64_t)height_ * (uint64_t)width_;
    const bool root = rank_ == dst_rank;
    std::unique_ptr<Grid> grid;
    detail::Buffer landing;
    // bands per round: up to 8 travel in one group; the landing area on the root is kept below ~4 GB
    const int per_round = (int)std::max<uint64_t>(1, std::min<uint64_t>(PCR_HIP_MAX_XFER_ARRAYS, ((uint64_t)4 << 30) / std::max<uint64_t>(whole * 4, 1)));
    Status local = Status::success();           // a local failure is announced through the transfer's own agreement (null arrays)
    if (root) {
        std::vector<BandDesc> descs;
        for (int b = 0; b < nb; ++b) descs.push_back(part->band_desc(b));
        grid = Grid::create(width_, height_, descs, MemoryLocation::Host);
        if (!grid) local = Status::error(StatusCode::OutOfMemory, "ShardedPipeline::gather: host allocation of the whole grid failed");
        else local = landing.allocate((size_t)std::min(per_round, nb) * whole * sizeof(float), MemoryLocation::Device);
    }
    for (int b0 = 0; b0 < nb; b0 += per_round) {
        const int k = std::min(per_round, nb - b0);
        const void* send[PCR_HIP_MAX_XFER_ARRAYS] = {};
        void* recv[PCR_HIP_MAX_XFER_ARRAYS] = {};
        int32_t elem[PCR_HIP_MAX_XFER_ARRAYS] = {};
        for (int a = 0; a < k; ++a) {
            send[a] = pipe_->result_band_device(b0 + a);
            recv[a] = root && local.ok() ? static_cast<float*>(landing.data()) + (size_t)a * whole : nullptr;
            elem[a] = 4;
        }
        uint64_t counts[PCR_HIP_MAX_ROUTE_PARTS] = {};
        Status s = detail::hip_status(pcr_hip_comm_gatherv(comm_, k, send, recv, elem, strip, root ? whole : 0, counts, dst_rank, stream));
        if (!s.ok()) return local.ok() ? s : local;
        if (!root) continue;
        uint64_t got = 0;
        for (int p = 0; p < world_; ++p) got += counts[p];
        if (got != whole)
            return Status::error(StatusCode::InvalidArgument, "ShardedPipeline::gather: the ranks' strips hold " + std::to_string((unsigned long long)got) +
                                 " cells, the grid has " + std::to_string((unsigned long long)whole));
        for (int a = 0; a < k; ++a) {
            s = detail::hip_status(pcr_hip_memcpy_d2h(grid->band_f32(b0 + a), recv[a], (size_t)whole * sizeof(float), stream));
            if (!s.ok()) return s;
        }
        if (!(s = detail::hip_status(pcr_hip_stream_synchronize(stream))).ok()) return s;     // (the landing area is reused)
    }
    if (root) *out = std::move(grid);
    return Status::success();
}

Status ShardedPipeline::ingest_unrouted(const PointCloud& cloud_in, size_t* ingested) {
    if (ingested) *ingested = 0;
    void* stream = pipe_->stream_handle();
    // A rank whose own part fails still takes part in every collective of the round (with nothing to send), so that the
    // others are not left waiting; its error is returned at the end.
    Status local = Status::success();
    std::unique_ptr<PointCloud> staged;
    const PointCloud* cloud = &cloud_in;
    if (cloud->count() > 0 && cloud->location() != MemoryLocation::Device) {
        staged = cloud->to(MemoryLocation::Device);
        if (!staged) local = Status::error(StatusCode::OutOfMemory, "ShardedPipeline::ingest_unrouted: cannot copy the cloud to the device");
        else cloud = staged.get();
    }
    const std::vector<std::string> names = cloud->channel_names();
    const int narrays = 2 + (int)names.size();
    if (narrays > PCR_HIP_MAX_ROUTE_ARRAYS)
        return Status::error(StatusCode::InvalidArgument, "ShardedPipeline::ingest_unrouted: at most six channels are routed in one pass");
    int32_t elem[PCR_HIP_MAX_ROUTE_ARRAYS] = {8, 8};
    for (size_t c = 0; c < names.size(); ++c) {
        elem[2 + c] = (int32_t)data_type_size(cloud->channel(names[c])->dtype);
        if (elem[2 + c] != 4 && elem[2 + c] != 8)
            return Status::error(StatusCode::InvalidArgument, "ShardedPipeline::ingest_unrouted: channel " + names[c] + " is not 4 or 8 bytes wide");
    }
    const uint64_t n = local.ok() ? cloud->count() : 0;
    std::vector<int32_t> splits((size_t)world_ + 1);
    for (int r = 0; r < world_; ++r) splits[(size_t)r] = row_block(r, world_, height_, align_).first;
    splits[(size_t)world_] = height_;
    pcr_hip_grid hg{};
    hg.min_x = grid_.bounds.min_x; hg.min_y = grid_.bounds.min_y; hg.max_x = grid_.bounds.max_x; hg.max_y = grid_.bounds.max_y;
    hg.cell_size_x = grid_.cell_size_x; hg.cell_size_y = grid_.cell_size_y;
    hg.width = grid_.width; hg.height = grid_.height; hg.tile_width = grid_.tile_width; hg.tile_height = grid_.tile_height;
    hg.own_row0 = 0; hg.own_row1 = grid_.height; hg.state_row0 = 0; hg.state_rows = grid_.height;    // the routing sees the whole grid

    // 1. owner of every point, points per owner
    std::vector<uint64_t> send_counts((size_t)PCR_HIP_MAX_ROUTE_PARTS, 0), recv_counts((size_t)PCR_HIP_MAX_ROUTE_PARTS, 0);
    detail::Buffer dest, counts, grouped[PCR_HIP_MAX_ROUTE_ARRAYS];
    if (n > 0 && local.ok()) {
        local = dest.allocate((size_t)n, MemoryLocation::Device);
        if (local.ok()) local = counts.allocate(2 * (size_t)world_ * sizeof(uint64_t), MemoryLocation::Device);
        if (local.ok())
            local = detail::hip_status(pcr_hip_route_count(&hg, splits.data(), world_, cloud->x(), cloud->y(), nullptr, n,
                                                           static_cast<uint8_t*>(dest.data()), static_cast<unsigned long long*>(counts.data()), stream));
        if (local.ok()) local = detail::hip_status(pcr_hip_memcpy_d2h(send_counts.data(), counts.data(), (size_t)world_ * sizeof(uint64_t), stream));
        if (local.ok()) local = detail::hip_status(pcr_hip_stream_synchronize(stream));
        if (!local.ok()) std::fill(send_counts.begin(), send_counts.end(), 0);
    }
    uint64_t total_send = 0;
    std::vector<uint64_t> cursors((size_t)world_);
    for (int p = 0; p < world_; ++p) { cursors[(size_t)p] = total_send; total_send += send_counts[(size_t)p]; }
    // 2. every array regrouped by owner
    const void* src[PCR_HIP_MAX_ROUTE_ARRAYS] = {};
    void* grp[PCR_HIP_MAX_ROUTE_ARRAYS] = {};
    if (total_send > 0 && local.ok()) {
        src[0] = cloud->x();
        src[1] = cloud->y();
        for (size_t c = 0; c < names.size(); ++c) src[2 + c] = cloud->channel_data(names[c]);
        for (int a = 0; a < narrays && local.ok(); ++a) {
            local = grouped[a].allocate((size_t)total_send * (size_t)elem[a], MemoryLocation::Device);
            grp[a] = grouped[a].data();
        }
        unsigned long long* d_cursors = static_cast<unsigned long long*>(counts.data()) + world_;
        if (local.ok()) local = detail::hip_status(pcr_hip_memcpy_h2d(d_cursors, cursors.data(), (size_t)world_ * sizeof(uint64_t), stream));
        if (local.ok())
            local = detail::hip_status(pcr_hip_route_scatter(static_cast<const uint8_t*>(dest.data()), n, world_, d_cursors, narrays, src, grp, elem, stream));
        if (!local.ok()) { std::fill(send_counts.begin(), send_counts.end(), 0); total_send = 0; }
    }
    // 3. the groups travel to their owners
    uint64_t total_recv = 0;
    if (world_ > 1) {
        Status s = detail::hip_status(pcr_hip_comm_alltoall_counts(comm_, send_counts.data(), recv_counts.data(), stream));
        if (!s.ok()) return local.ok() ? s : local;
        for (int p = 0; p < world_; ++p) total_recv += recv_counts[(size_t)p];
    } else {
        total_recv = total_send;
    }
    std::unique_ptr<PointCloud> mine = PointCloud::create((size_t)std::max<uint64_t>(total_recv, 1), MemoryLocation::Device);
    void* dst[PCR_HIP_MAX_ROUTE_ARRAYS] = {};
    bool have = mine != nullptr;
    if (have) {
        for (size_t c = 0; c < names.size() && have; ++c) have = mine->add_channel(names[c], cloud->channel(names[c])->dtype).ok();
        if (have) have = mine->resize((size_t)total_recv).ok();
    }
    if (have) {
        dst[0] = mine->x();
        dst[1] = mine->y();
        for (size_t c = 0; c < names.size(); ++c) dst[2 + c] = mine->channel_data(names[c]);
    } else if (local.ok()) {
        local = Status::error(StatusCode::OutOfMemory, "ShardedPipeline::ingest_unrouted: cannot allocate the routed cloud on the device");
    }
    if (world_ > 1) {
        // (a rank without a landing area announces no room: every rank refuses the round together)
        Status s = detail::hip_status(pcr_hip_comm_alltoallv(comm_, narrays, grp, dst, elem, send_counts.data(), have ? total_recv : 0,
                                                             nullptr, stream));
        if (!s.ok()) return local.ok() ? s : local;
    } else if (have && total_recv > 0 && local.ok()) {
        for (int a = 0; a < narrays && local.ok(); ++a)
            local = detail::hip_status(pcr_hip_memcpy_d2d(dst[a], grp[a], (size_t)total_recv * (size_t)elem[a], stream));
    }
    // 4. ingest what arrived (ingest() agrees on the Line reach over the ranks first; a rank that failed locally still joins
    //    that agreement, with an empty cloud)
    Status s = Status::success();
    if (local.ok() && have) {
        s = ingest(*mine);
        if (s.ok() && ingested) *ingested = (size_t)total_recv;
    } else {
        std::unique_ptr<PointCloud> none = PointCloud::create(1, MemoryLocation::Device);
        if (none) { (void)none->resize(0); (void)ingest(*none); }
    }
    Status sync = pipe_->synchronize();                      // `mine` and the staging buffers are freed on return
    if (!local.ok()) return local;
    return s.ok() ? sync : s;
}

uint64_t ShardedPipeline::bytes_sent() const {
    uint64_t b = 0;
    if (comm_) pcr_hip_comm_stats(comm_, nullptr, &b);
    return b;
}

}  // namespace pcr
