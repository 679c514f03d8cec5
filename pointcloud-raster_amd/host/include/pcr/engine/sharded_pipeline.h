// sharded_pipeline.h -- one grid over the GPUs of a node, from C++: Pipeline on this rank's row block + the native
// halo exchange (pcr_hip_comm_*: RCCL send/recv to rank +- 1 over xGMI, include/pcr_hip.h).
//
// The reference's Pipeline is single-device (include/pcr/engine/pipeline.h:68 `cuda_device_id`); this class is new
// work at the same API level (create / ingest / finalize / result), so that a C++ caller can shard without Python:
//
//     uint8_t id[ShardedPipeline::kIdBytes];
//     if (rank == 0) ShardedPipeline::make_id(id);
//     /* carry id to every rank (MPI_Bcast, a file, ...) */
//     auto sp = ShardedPipeline::create(cfg /* WHOLE grid */, id, rank, world, device);
//     sp->ingest(cloud);   // any superset of the points whose centre row this rank owns
//     sp->ingest_unrouted(shard);   // or: an ARBITRARY shard of the cloud -- its points travel to their owners first
//     sp->finalize();      // halo reduce + touched-tile union + local finalize (+ ONE GeoTIFF from rank 0 when
//                          // cfg.output_path is set, as the reference writes one file: src/engine/pipeline.cpp:1351-1361)
//     sp->result();        // rows [row_begin(), row_end()) of every band: this rank's STRIP
//     sp->gather(0, &grid);  // the whole grid on one rank (the strips in rank order = row order)
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "pcr/engine/pipeline.h"

struct pcr_hip_comm;          // include/pcr_hip.h

namespace pcr {

class ShardedPipeline {
public:
    static constexpr int kIdBytes = 128;
    ~ShardedPipeline();
    ShardedPipeline(const ShardedPipeline&) = delete;
    ShardedPipeline& operator=(const ShardedPipeline&) = delete;

    /// Rank 0: the communicator's bootstrap id (ncclUniqueId).
    static Status make_id(uint8_t* id128);
    /// Rows [r0, r1) of `rank`: contiguous, balanced, edges multiples of `align` (align = tile height: no exchange).
    static std::pair<int, int> row_block(int rank, int world, int height, int align = 1);
    /// Collective over the `world` ranks.  nullptr on failure (see create_error()).
    static std::unique_ptr<ShardedPipeline> create(PipelineConfig cfg, const uint8_t* id128, int rank, int world,
                                                   int device, int align = 1);
    static const std::string& create_error();

    Status ingest(const PointCloud& cloud);
    /// SURVEY section 8e "device-side partition + peer copy": `cloud` is an ARBITRARY shard of the whole cloud (a file chunk
    /// per rank).  Its points are grouped by owner on the device (pcr_hip_route_count / _scatter), travel to their owners
    /// (pcr_hip_comm_alltoallv: counts agreed first, one grouped ncclSend / ncclRecv round for x, y and every channel), and
    /// each rank ingests what it receives.  Collective: every rank calls it once per round, with an empty cloud (count() == 0)
    /// if it has nothing to contribute.  Channels must be 4 or 8 bytes wide; at most six of them.  `ingested` = points received.
    Status ingest_unrouted(const PointCloud& cloud, size_t* ingested = nullptr);
    /// The exchange alone (finalize() calls it): apron rows to their owners, touched-tile union.
    Status exchange();
    Status finalize();
    /// `.pcrt` checkpoints (collective).  save_state: the exchange first; with blocks of whole reference-tile rows every rank
    /// writes the tiles it owns (their union is the checkpoint), with blocks that cut tiles the planes' owned rows are gathered
    /// to rank 0, which writes the whole grid's tiles -- either way the files an unsharded pipeline would write.  load_state:
    /// every rank takes its own rows out of the tile files they meet (also PipelineConfig::resume at create).
    Status save_state(const std::string& dir = "");
    Status load_state(const std::string& dir = "") { return pipe_->load_state(dir); }
    const Grid* result() const { return pipe_->result(); }
    /// Collective, after finalize(): every rank's strip to `dst_rank` (pcr_hip_comm_gatherv, device to device), where *out
    /// becomes a host Grid of the WHOLE grid (width x height, the strips' bands); *out stays null on the other ranks.  The
    /// reference's result() is one grid and it writes one file (src/engine/pipeline.cpp:1175-1186, 1351-1361).
    Status gather(int dst_rank, std::unique_ptr<Grid>* out);
    Pipeline& pipeline() { return *pipe_; }
    int row_begin() const { return r0_; }
    int row_end() const { return r1_; }
    int halo_rows() const { return halo_; }
    bool tiles_local() const { return tiles_local_; }      // every block edge on a reference-tile row: nothing to exchange
    uint64_t bytes_sent() const;

private:
    ShardedPipeline() = default;
    std::unique_ptr<Pipeline> pipe_;
    ::pcr_hip_comm* comm_ = nullptr;
    int rank_ = 0, world_ = 1, r0_ = 0, r1_ = 0, halo_ = 0, width_ = 0, height_ = 0, align_ = 1;
    GridConfig grid_;                 // the WHOLE grid
    std::string output_path_;         // taken from the configuration: rank 0 writes ONE GeoTIFF at finalize()
    std::string state_dir_;
    std::vector<ReductionSpec> reductions_;
    bool tiles_local_ = false;
    bool line_hl_groups_ = false;     // a Line group with a per-point half_length channel: ingest agrees on its reach first
};

}  // namespace pcr
