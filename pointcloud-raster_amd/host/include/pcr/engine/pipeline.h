// pcr/engine/pipeline.h -- the public engine API (drop-in for the reference's
// include/pcr/engine/pipeline.h:20-145): ReductionSpec, ExecutionMode, PipelineConfig,
// ProgressInfo, Pipeline{create, validate, ingest, finalize, run, set_progress_callback,
// result, stats}.  Behind it: the MI355X HIP engine (include/pcr_hip.h) for GPU / Auto / Hybrid,
// and a host engine for ExecutionMode::CPU and for the cases in which the reference falls back to
// its CPU mode (no device + gpu_fallback_to_cpu, Auto without a GPU: src/engine/pipeline.cpp:100-131)
// -- announced on stderr with the reference's own Warning / Info line, visible through engine(),
// and forbidden altogether by PCR_REQUIRE_GPU_ENGINE=1.
#pragma once

#include "pcr/core/grid_config.h"
#include "pcr/core/types.h"
#include "pcr/engine/filter.h"
#include "pcr/engine/glyph.h"

#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace pcr {

class PointCloud;
class Grid;

struct ReductionSpec {
    std::string value_channel;
    ReductionType type = ReductionType::Sum;
    std::string weight_channel;        // declared; WeightedAverage uses weight 1 with the Point glyph (reference behaviour)
    std::string timestamp_channel;
    float percentile = 0.5f;
    std::string output_band_name;      // default "{value_channel}_{int(type)}"
    GlyphSpec glyph;
};

enum class ExecutionMode : uint8_t { CPU, GPU, Auto, Hybrid };

struct PipelineConfig {
    GridConfig grid;
    std::vector<ReductionSpec> reductions;
    FilterSpec filter;

    CRS target_crs;
    bool auto_reproject = true;

    ExecutionMode exec_mode = ExecutionMode::Auto;   // GPU, Auto and Hybrid run the HIP engine; CPU the host engine

    // A grid whose accumulation state (4 B per cell and plane, + the finalized bands) exceeds gpu_memory_budget (0 = ~80 %
    // of the free GPU memory) is processed OUT OF CORE: in row bands of whole reference-tile rows, one band's state in HBM at
    // a time, the others in host memory up to host_cache_budget (0 = ~50 % of the free host memory), beyond that in files
    // under state_dir (a temporary directory when empty) -- the role of the reference's TileManager LRU + disk spill
    // (src/engine/tile_manager.cpp:76-375).
    size_t gpu_memory_budget = 0;
    size_t host_cache_budget = 0;
    size_t chunk_size = 0;

    size_t gpu_pool_size_bytes = 512 * 1024 * 1024;  // initial scratch arena of the engine
    int cuda_device_id = 0;                          // HIP device ordinal
    bool use_cuda_streams = true;
    bool gpu_fallback_to_cpu = true;                 // GPU mode without a usable device continues on the host engine (with a Warning)
    bool gpu_require_strict = false;

    size_t cpu_threads = 0;                          // host engine: OpenMP threads (0 = all cores)
    size_t hybrid_cpu_threads = 0;

    std::string state_dir;
    bool resume = false;

    std::string output_path;                         // GeoTIFF writing is not part of this build
    bool write_cog = false;

    // ---- extensions (not in the reference; defaults keep reference behaviour) ----------
    MemoryLocation result_location = MemoryLocation::Host;   // Device: finalize() leaves bands in HBM
    // Row-block sharding of the grid over GPUs: this pipeline ingests points whose centre row
    // is in [shard_row_begin, shard_row_end) and finalizes those rows; -1 = whole grid.
    int shard_row_begin = -1;
    int shard_row_end = -1;
    // Rows of state kept beyond each side of the owned block (glyph footprints of points near the block edge land
    // there and are sent to the owning rank).  -1: sized from the glyph defaults (Gaussian: max_radius_cells; Line:
    // default_half_length / |cell_size_y|, which max_radius_cells does NOT cap on north-up grids).  A Line glyph with a
    // per-point half_length channel can reach further: ingest() checks every cloud and refuses (InvalidArgument,
    // naming the rows needed) rather than clip silently -- raise this knob then.  Same value on every rank.
    int shard_halo_rows = -1;
    int scatter_path = 0;                            // 0 auto, 1 direct atomics, 2 binned LDS tiles, 3 moments+convolution (Gaussian)
    // The first Point scatter of a pipeline also stores the finished bands (they are dropped again by any later ingest): the
    // right default when finalize() follows a single ingest; pipelines that always ingest several clouds save one wasted band
    // store by switching it off.
    bool finalize_with_first_ingest = true;
};

struct ProgressInfo {
    size_t collections_processed = 0;
    size_t collections_total = 0;
    size_t points_processed = 0;
    size_t tiles_active = 0;
    float elapsed_seconds = 0.0f;
};

using ProgressCallback = std::function<bool(const ProgressInfo& info)>;

class Pipeline {
public:
    ~Pipeline();

    static std::unique_ptr<Pipeline> create(const PipelineConfig& config);   // nullptr on failure

    Status validate() const;
    Status ingest(const PointCloud& cloud);
    /// Extension: like ingest, but returns once the work is enqueued when the cloud is
    /// page-locked (HostPinned) or device-resident; the cloud must stay untouched until synchronize() /
    /// finalize().  Pageable host clouds behave as in ingest.
    Status ingest_async(const PointCloud& cloud);
    /// Extension: streams a PCRP / CSV file through two page-locked chunk buffers -- the file read of chunk
    /// k+1 overlaps the host-to-device copy and the kernels of chunk k.  `points_read` (optional) = rows read.
    Status ingest_file(const std::string& path, size_t chunk_points = 4u << 20, size_t* points_read = nullptr);
    Status finalize();
    /// Extension, the counterpart of ingest_async: with a device-resident result (result_location = Device, no output_path)
    /// the finalize kernels are only ENQUEUED on the pipeline's stream -- result() is valid at once, its bands are complete
    /// after synchronize() (or for anything ordered after it on stream_handle()).  Back-to-back pipelines then never leave
    /// the device idle for a host round trip.  Any other configuration behaves as finalize().
    Status finalize_async();
    Status run(const std::vector<const PointCloud*>& clouds);
    void set_progress_callback(ProgressCallback cb);
    const Grid* result() const;
    ProgressInfo stats() const;

    // ---- checkpoint / resume in the reference's `.pcrt` tile-state format (pcr/io/tile_state_io.h).
    // save_state writes one file per touched reference tile and ReductionSpec (directly into `dir` for a
    // single reduction -- the reference's layout -- else into dir/reduction_<i>/); load_state sets the
    // device state from such files (also ones written by the reference) and marks their tiles touched.
    // An empty `dir` means PipelineConfig::state_dir.  PipelineConfig::resume = true loads at create().
    // Unlike the reference, finalize() never writes state files implicitly.
    Status save_state(const std::string& dir = "");
    Status load_state(const std::string& dir = "");

    // ---- extensions for row-block sharded runs (halo exchange is driven by the caller,
    //      e.g. torch.distributed over RCCL: see pcr/distributed.py) -------------------------
    struct PlaneView {
        void* device_ptr;      // first float of the plane (row state_row_begin)
        int plane_kind;        // PCR_HIP_PLANE_* of include/pcr_hip.h
        int group;             // accumulation group index
        int reach_rows;        // rows the group's glyph can reach beyond a point's centre row (0: Point glyph -- its halo
                               // rows never receive anything and need no exchange)
    };
    int halo_rows() const;                       // rows kept above/below the owned block
    // Row-block shards: rows the Line groups need beyond a centre row for this cloud (0: no check applies).  ingest()
    // refuses a cloud that needs more than halo_rows(); sharded callers agree on MAX over ranks first.
    Status line_reach_rows(const PointCloud& cloud, int* rows);
    int state_row_begin() const;
    int state_row_count() const;
    std::vector<PlaneView> state_planes() const;
    /// Accumulation group of every ReductionSpec, in order (the `group` of state_planes()); empty on the other engines.
    std::vector<int> reduction_groups() const;
    void* tile_touched_device(int* tiles_x, int* tiles_y) const;
    /// The same flags for READING only (nothing is assumed to change: bands a scatter stored stay valid).
    const void* tile_touched_device_readonly(int* tiles_x, int* tiles_y) const;
    /// Row-block shards: ORs `d_union` (tiles_x * tiles_y device words: the all-reduced flags of every rank) into this
    /// pipeline's flags on its stream; bands a scatter stored are dropped -- on the device -- only when a flag really changed.
    Status merge_touched(const void* d_union);
    /// The finished band `band` of the last finalize() in DEVICE memory (own rows x width floats; the device-resident
    /// result's band, or the device-side copy a host-resident result was made from): what a sharded gather sends.
    /// nullptr before the first finalize, out of core, or for a band index outside the result.
    const float* result_band_device(int band) const;
    Status synchronize();
    void* stream_handle() const;                 // the hipStream_t every kernel of this pipeline runs on (may be null)
    // per-kernel HIP-event timing of the scatter kernels (roofline reporting)
    struct KernelTime { std::string name; unsigned launches; double total_ms; };
    void profile_enable(bool on, const std::string& only_kernel = "");   // only_kernel: bracket just that kernel with events
    std::vector<KernelTime> profile_read(bool reset);
    // path and LDS tiling the last scatter used (0 direct, 1 binned), exact valid-point count
    // bands_with_scatter: accumulation groups whose finished bands the scatter that defined their planes was asked to store
    // too and that nothing has invalidated since (finalize skips their kernel when the device confirms)
    struct ScatterInfo { int path; int lds_tile_w, lds_tile_h, lds_apron, num_bins; size_t points_in, points_valid; int scatter_chunk; int bands_with_scatter; };
    ScatterInfo last_scatter() const;

    /// True when the grid's state did not fit the device budget and the pipeline sweeps it in row bands (out of core).
    bool out_of_core() const;
    /// Out of core: the pipeline's own directory of evicted bands -- `.pcrt` tile files in the reference's layout, removed with
    /// the pipeline (save_state() writes checkpoints that stay).  Empty otherwise.
    std::string spill_dir() const;
    /// "hip" (the MI355X engine) or "host" (ExecutionMode::CPU, or a fallback the reference would have taken too).
    const char* engine() const;
    /// OpenMP threads of the host engine (cpu_threads, or every CPU the process may use: affinity and cgroup quota); 0 on "hip".
    int host_threads() const;

private:
    Pipeline() = default;
    struct Impl;
    struct Banded;                       // out-of-core driver: row bands of whole reference-tile rows, one in HBM at a time
    struct Host;                         // the pipeline on the host engine (host/src/host_pipeline.h)
    std::unique_ptr<Impl> impl_;
    std::unique_ptr<Banded> banded_;
    std::unique_ptr<Host> host_;
};

// Why the last Pipeline::create() on this thread returned nullptr.
const std::string& pipeline_create_error();

}  // namespace pcr
