// sharded_pipeline.cpp -- see pcr/engine/sharded_pipeline.h.
#include "pcr/engine/sharded_pipeline.h"

#include <algorithm>
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
    // (a per-point half_length Line group can need more halo than the shard keeps: Pipeline::ingest refuses such a
    //  cloud before anything is accumulated; callers that must refuse TOGETHER reduce Pipeline::line_reach_rows first)
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
