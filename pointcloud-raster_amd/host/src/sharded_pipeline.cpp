// sharded_pipeline.cpp -- see pcr/engine/sharded_pipeline.h.
#include "pcr/engine/sharded_pipeline.h"

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "buffer.h"
#include "pcr_hip.h"

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
    return pipe_->finalize();
}

uint64_t ShardedPipeline::bytes_sent() const {
    uint64_t b = 0;
    if (comm_) pcr_hip_comm_stats(comm_, nullptr, &b);
    return b;
}

}  // namespace pcr
