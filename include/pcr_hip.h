/*
 * pcr_hip.h -- C-ABI of libpcr_hip.so: the MI355X (gfx950) engine behind
 * pcr::Pipeline::ingest / finalize.
 *
 * Plain C: opaque handles, raw host/device pointers and sizes; no C++ or torch types.
 * Every function returns a pcr_hip_status (same numbering as the reference's
 * pcr::StatusCode, include/pcr/core/types.h:118-126; HIP failures map to
 * PCR_HIP_CUDA_ERROR to keep the reference's code name); the message of the last
 * failure on the calling thread is pcr_hip_last_error().
 *
 * The reference has no C ABI of its own.  Each entry point replaces one of its C++
 * seams, cited as "replaces:" (file:line into BigHippo123/pointcloud-raster).
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Layout conventions
 *   - points: SoA, x/y float64, channels float32 (include/pcr/core/point_cloud.h:29-103)
 *   - state:  band-sequential float32 planes in GRID layout, plane f at
 *             base + f*plane_stride, cell (row, col) at (row - state_row0)*width + col
 *             (the reference keeps per-tile planes, include/pcr/ops/reduction_op.h:45;
 *             tiles here are only a semantic: clip rectangle + "touched" flag)
 *   - all pointers named d_* are device pointers, h_* host pointers.
 */
#ifndef PCR_HIP_H
#define PCR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCR_HIP_ABI_VERSION 5   /* 3 (round 4): + comm_halo_plan / comm_agree_max_i32 / signed_max_f32_masked / copy_kernel; planes_fresh takes 0, 1, 2;
                                 * 4: + engine_finalize_with_scatter / engine_finalize_taken / finalize_group_unless / touched_union;
                                 * 5 (round 5): + comm_alltoall_counts / comm_alltoallv / comm_gatherv / comm_xfer_plan / touched_union_owned; halo_reduce resets the sent apron rows */

typedef enum pcr_hip_status {
    PCR_HIP_OK = 0,
    PCR_HIP_INVALID_ARGUMENT = 1,
    PCR_HIP_OUT_OF_MEMORY = 2,
    PCR_HIP_CUDA_ERROR = 3,        /* a HIP runtime failure (name kept from the reference) */
    PCR_HIP_IO_ERROR = 4,
    PCR_HIP_CRS_ERROR = 5,
    PCR_HIP_NOT_IMPLEMENTED = 6
} pcr_hip_status;

/* pcr::ReductionType numbering (types.h:33-45); only these six are registered
 * (src/ops/reduction_registry.cpp:173-184). */
enum {
    PCR_HIP_SUM = 0, PCR_HIP_MAX = 1, PCR_HIP_MIN = 2,
    PCR_HIP_AVERAGE = 3, PCR_HIP_WEIGHTED_AVERAGE = 4, PCR_HIP_COUNT = 5
};

/* pcr::GlyphType numbering (include/pcr/engine/glyph.h:11-15). */
enum { PCR_HIP_GLYPH_POINT = 0, PCR_HIP_GLYPH_LINE = 1, PCR_HIP_GLYPH_GAUSSIAN = 2 };

/* Accumulation planes a scatter can feed in one pass over the points. */
enum {
    PCR_HIP_PLANE_SUM = 1u,   /* += value * w        (Sum; numerator of Average/WeightedAverage) */
    PCR_HIP_PLANE_WGT = 2u,   /* += w  (w = 1 for the Point glyph: Count; denominator)           */
    PCR_HIP_PLANE_MAX = 4u,   /* fmaxf, identity -FLT_MAX (Point glyph only)                      */
    PCR_HIP_PLANE_MIN = 8u    /* fminf, identity +FLT_MAX (Point glyph only)                      */
};

typedef void* pcr_hip_stream;      /* hipStream_t; NULL = the null stream */

/* The fields of pcr::GridConfig the hot path reads (include/pcr/core/grid_config.h:17-38)
 * plus the row window a device holds when the grid is row-block sharded over GPUs. */
typedef struct pcr_hip_grid {
    double min_x, min_y, max_x, max_y;   /* bounds; origin is (min_x, max_y) */
    double cell_size_x, cell_size_y;     /* cell_size_y < 0 on north-up grids */
    int32_t width, height;               /* whole grid, cells */
    int32_t tile_width, tile_height;     /* reference tiling (clip + touched semantics) */
    int32_t own_row0, own_row1;          /* this device ingests points whose centre row is in [own_row0, own_row1) */
    int32_t state_row0, state_rows;      /* rows held in the state planes: [state_row0, state_row0 + state_rows) */
} pcr_hip_grid;

/* Device planes of one accumulation group.  A NULL plane is not accumulated. */
typedef struct pcr_hip_planes {
    float* d_sum;
    float* d_wgt;
    float* d_max;
    float* d_min;
} pcr_hip_planes;

/* pcr::GlyphSpec numeric fields (glyph.h:20-42) + per-point channel arrays (device, may be NULL). */
typedef struct pcr_hip_glyph {
    int32_t type;
    float default_direction, default_half_length;
    float default_sigma_x, default_sigma_y, default_rotation;
    float max_radius_cells;
    const float* d_direction;
    const float* d_half_length;
    const float* d_sigma_x;
    const float* d_sigma_y;
    const float* d_rotation;
} pcr_hip_glyph;

/* Counters of the last scatter on an engine (diagnostics; exact). */
typedef struct pcr_hip_scatter_stats {
    uint64_t points_in;        /* points offered */
    uint64_t points_valid;     /* inside bounds and inside [own_row0, own_row1) */
    int32_t path;              /* 0 = direct global atomics, 1 = binned LDS tiles, 2 = moments + convolution */
    int32_t lds_tile_w, lds_tile_h, lds_apron, num_bins;
    int32_t scatter_chunk;     /* binned path: points per workgroup of the record-scatter pass (0 otherwise) */
    int32_t reserved_;
} pcr_hip_scatter_stats;

const char* pcr_hip_last_error(void);
int pcr_hip_abi_version(void);

/* ---- devices.  replaces: cuda_device_available/count/name/get_memory_info,
 *      include/pcr/core/types.h:156-219, and cudaSetDevice in src/engine/pipeline.cpp:133-160 */
int pcr_hip_device_count(int* count);
int pcr_hip_set_device(int device_id);
int pcr_hip_get_device(int* device_id);
int pcr_hip_device_name(int device_id, char* buf, size_t buf_len);
int pcr_hip_mem_info(size_t* free_bytes, size_t* total_bytes);
int pcr_hip_device_synchronize(void);

/* ---- streams / events.  replaces: the single cudaStream of src/engine/pipeline.cpp:198-213 */
int pcr_hip_stream_create(pcr_hip_stream* out);
int pcr_hip_stream_destroy(pcr_hip_stream s);
int pcr_hip_stream_synchronize(pcr_hip_stream s);
int pcr_hip_event_create(void** out);
int pcr_hip_event_destroy(void* ev);
int pcr_hip_event_record(void* ev, pcr_hip_stream s);
int pcr_hip_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);   /* synchronizes on ev_stop */

/* ---- memory.  replaces: cudaMalloc/cudaMallocHost/cudaMemcpy* in src/core/point_cloud.cpp:55-73,
 *      382-512 (PointCloud::to / to_device_async) and src/core/grid.cpp device paths */
int pcr_hip_malloc(void** d_ptr, size_t bytes);
int pcr_hip_free(void* d_ptr);
int pcr_hip_host_alloc(void** h_ptr, size_t bytes);          /* pinned */
int pcr_hip_host_free(void* h_ptr);
int pcr_hip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, pcr_hip_stream s);  /* async when h_src is pinned */
int pcr_hip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, pcr_hip_stream s);
int pcr_hip_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, pcr_hip_stream s);
/* Device-to-device copy by a hand-written float4 kernel (16-byte aligned pointers and size; nontemporal != 0: loads and
 * stores that bypass the caches' allocation).  The yardstick bench.py reports as `measured_copy_GBps`: the streaming rate
 * this box really reaches, next to the 8 TB/s data-sheet peak (no reference counterpart: measurement only). */
int pcr_hip_copy_kernel(void* d_dst, const void* d_src, size_t bytes, int nontemporal, pcr_hip_stream s);
int pcr_hip_memset(void* d_ptr, int byte_value, size_t bytes, pcr_hip_stream s);

/* ---- arena.  replaces: pcr::MemoryPool (include/pcr/engine/memory_pool.h:18-50,
 *      src/engine/memory_pool.cu:24-59): bump allocator over one device allocation, 256-B aligned */
typedef struct pcr_hip_arena pcr_hip_arena;
int pcr_hip_arena_create(pcr_hip_arena** out, size_t bytes);
int pcr_hip_arena_destroy(pcr_hip_arena* a);
int pcr_hip_arena_alloc(pcr_hip_arena* a, size_t bytes, void** d_ptr);
int pcr_hip_arena_reset(pcr_hip_arena* a);
int pcr_hip_arena_stats(const pcr_hip_arena* a, size_t* capacity, size_t* used, size_t* high_water);
/* The scatter engine's own scratch (routing keys, records, moment planes) is one such arena per device, shared by
 * every engine on it, grow-only, borrowed exclusively for the duration of a scatter's enqueue (csrc/engine.hip):
 * capacity and high-water mark in bytes, number of borrows and of (re)allocations since process start. */
int pcr_hip_device_scratch_stats(int device, size_t* capacity, size_t* high_water, uint64_t* borrows, uint64_t* grows);

/* ---- state planes.  replaces: init_tile_state / merge_tile_state / finalize_tile,
 *      include/pcr/engine/grid_merge.h:22-41 (src/engine/grid_merge.cu:116-183), and the CPU
 *      finalize loop of src/engine/pipeline.cpp:1204-1286 (NaN fill, untouched tiles skipped) */
int pcr_hip_state_floats(int rtype, int* k);                                   /* reduction_registry.cpp:197-208 */
int pcr_hip_plane_fill(float* d_plane, float value, int64_t cells, pcr_hip_stream s);
int pcr_hip_state_init(int rtype, float* d_state, int64_t cells, pcr_hip_stream s);         /* K planes, stride = cells */
int pcr_hip_state_merge(int rtype, float* d_dst, const float* d_src, int64_t cells, pcr_hip_stream s);
/* merge one plane with the op of its PCR_HIP_PLANE_* kind (add / fmaxf / fminf): halo rows, multi-ingest merges */
int pcr_hip_plane_merge(uint32_t plane_kind, float* d_dst, const float* d_src, int64_t cells, pcr_hip_stream s);
/* out[row, col] for rows [own_row0, own_row1): finalize(rtype) of the planes where the cell's
 * reference tile is touched, NaN elsewhere.  d_out holds (own_row1-own_row0)*width floats.
 * d_tile_touched: tiles_x*tiles_y words (NULL = all touched). */
int pcr_hip_finalize(int rtype, const pcr_hip_grid* g, const pcr_hip_planes* planes,
                     const uint32_t* d_tile_touched, float* d_out, pcr_hip_stream s);

/* Same, for up to PCR_HIP_MAX_FINALIZE_OUTPUTS reductions that read ONE group's planes: each plane is
 * read once and every band written once (Sum + Count + Average of one channel: 2 plane reads, 3 band
 * writes instead of 4 + 3). */
#define PCR_HIP_MAX_FINALIZE_OUTPUTS 8
int pcr_hip_finalize_group(const pcr_hip_grid* g, const pcr_hip_planes* planes, const uint32_t* d_tile_touched,
                           int n_out, const int* rtypes, float* const* d_outs, pcr_hip_stream s);
/* Same, but a no-op when the device word *d_bands_done is non-zero at the time the kernel runs (NULL: always runs):
 * the bands were already written by the scatter that defined the planes (pcr_hip_engine_finalize_with_scatter). */
int pcr_hip_finalize_group_unless(const pcr_hip_grid* g, const pcr_hip_planes* planes, const uint32_t* d_tile_touched,
                                  int n_out, const int* rtypes, float* const* d_outs, const uint32_t* d_bands_done,
                                  pcr_hip_stream s);

/* Union of another rank's touched-tile flags into this device's (row-block shards: a reference tile is touched if ANY rank saw
 * a point in it).  d_local[i] |= d_union[i] != 0, i < n; when that changed a flag, the n_words device words at d_bands_done
 * (may be NULL) are zeroed: bands that a scatter stored from the local flags alone are then stale and the finalize pass runs
 * (pcr_hip_finalize_group_unless).  Unchanged flags leave them alone -- the usual case on dense clouds. */
int pcr_hip_touched_union(uint32_t* d_local, const uint32_t* d_union, int32_t n, uint32_t* d_bands_done, int32_t n_words,
                          pcr_hip_stream s);
/* The same over a tiles_x x tiles_y flag grid for a device that owns rows of tile rows [own_tile_row0, own_tile_row1) only: a
 * flag that changes OUTSIDE that range is merged but leaves d_bands_done alone -- none of this device's bands depends on it
 * (non-tile-aligned row blocks: every other rank contributes tiles this one owns no row of). */
int pcr_hip_touched_union_owned(uint32_t* d_local, const uint32_t* d_union, int32_t tiles_x, int32_t tiles_y,
                                int32_t own_tile_row0, int32_t own_tile_row1, uint32_t* d_bands_done, int32_t n_words,
                                pcr_hip_stream s);

/* ---- scatter engine.  replaces: TileRouter::assign + sort + extract_batches
 *      (include/pcr/engine/tile_router_kernels.h:15-52, src/engine/tile_router.cpp:51-366),
 *      Accumulator::accumulate (include/pcr/engine/accumulator_kernels.h:13-23,
 *      src/engine/accumulator.cpp:33-59) and accumulate_glyph
 *      (include/pcr/engine/glyph_kernels.h:31-42), fused: no sort, no materialised indices. */
typedef struct pcr_hip_engine pcr_hip_engine;
/* scratch_bytes = 0: the engine grows its scratch arena on demand (the arena is shared by all engines of a device).
 * Test-only environment knobs, read here: PCR_HIP_DEBUG_MAX_BINS=<n> lowers the number of LDS tiles one binning pass
 * may count (8064) so that the large-grid paths (two-level sort, row bands) are reached on small grids;
 * PCR_HIP_DEBUG_TWO_LEVEL=0 forces the row-band sweep where the two-level sort would apply
 * (the passes on 16-byte glyph records count up to 16384 tiles: the knob applies to them as it stands).
 * PCR_HIP_TUNE_CONV=1|2 forces the vector-ALU | matrix-core column pass of the moment path whatever the radius (tests reach
 * both kernels on small shapes), PCR_HIP_CONV_WAVES=4|8 the workgroup shape of the matrix-core pass, PCR_HIP_CELL_TILE_H=<rows>
 * the height of the Gaussian cell tiles (read at scatter time): shapes only, results do not change.  PCR_HIP_RCCL=<path> names
 * the RCCL library pcr_hip_comm_* loads (default: the copy the process already has, else the one beside libamdhip64). */
int pcr_hip_engine_create(pcr_hip_engine** out, const pcr_hip_grid* g, size_t scratch_bytes, pcr_hip_stream s);
int pcr_hip_engine_destroy(pcr_hip_engine* e);
/* 0 = auto, 1 = force direct global atomics, 2 = force binned LDS tiles (INVALID_ARGUMENT if the grid cannot
 * be binned), 3 = force the separable moment + convolution path for Gaussians (other glyphs: as auto) */
int pcr_hip_engine_set_path(pcr_hip_engine* e, int path);
/* State of the planes handed to the NEXT pcr_hip_scatter_point / _glyph only (cleared by it).  Wrong hint = wrong results.
 *   0  they hold earlier contributions: the merge read-modify-writes (the default);
 *   1  every cell holds its identity value (just filled by pcr_hip_plane_fill / pcr_hip_state_init, nothing accumulated
 *      or loaded since): the Point tile-merge stores instead of read-modify-writing (saves one read of the planes);
 *   2  they are UNDEFINED (never filled): the scatter itself leaves every cell of the state window defined -- the binned
 *      Point path by storing every cell of every LDS tile, identity included (no pass of its own; a bin the scan had to
 *      split, or a window swept in several bands / two sort levels, falls back to filling first), the moment path by
 *      letting its row pass store instead of accumulate (one window over the whole state window; else it fills first),
 *      every other path by filling the planes with identity values before it accumulates (timed as `k_state_init`).
 * The reference initialises tile state inside ingest, on first acquire (src/engine/tile_manager.cpp:272-320,
 * src/engine/pipeline.cpp:688-691): with 2 this build's state initialisation is inside the ingest as well. */
int pcr_hip_engine_planes_fresh(pcr_hip_engine* e, int fresh);
/* Finalize fused into the scatter that defines the planes -- for the NEXT pcr_hip_scatter_point only (cleared by it).
 * When that scatter runs with planes_fresh = 2 on the binned path and its tile pass stores every cell of the state
 * window itself, the same pass also stores the finished bands (finalize(rtype) of the cell where its reference tile is
 * touched BY THIS SCATTER'S points, NaN elsewhere) from the LDS tile it has in hand: the planes are not read again
 * (Pipeline::finalize after a pipeline's only ingest: src/engine/pipeline.cpp:1154-1286 reads every tile's state back).
 * d_outs[i] holds own_rows * width floats, 16-byte aligned; the engine's owned rows must be its whole state window.
 * *d_bands_done (device word) is written by the scatter: 1 when every band cell was stored, 0 when the pass could not
 * (a bin the scan had to split).  pcr_hip_engine_finalize_taken: 1 when the last pcr_hip_scatter_point launched the
 * fused form at all (else *d_bands_done was not written and the bands are untouched).  The bands stay valid only while
 * nothing else changes the planes or the touched flags: the caller decides (pcr_hip_finalize_group_unless). */
int pcr_hip_engine_finalize_with_scatter(pcr_hip_engine* e, int n_out, const int* rtypes, float* const* d_outs,
                                         uint32_t* d_bands_done);
int pcr_hip_engine_finalize_taken(const pcr_hip_engine* e);
int pcr_hip_engine_stats(const pcr_hip_engine* e, pcr_hip_scatter_stats* out);
/* device array of tiles_x*tiles_y words, non-zero where a valid point's centre cell fell */
int pcr_hip_engine_tile_touched(pcr_hip_engine* e, uint32_t** d_tile_touched, int32_t* tiles_x, int32_t* tiles_y);

/* ---- point filter.  replaces: filter_points (include/pcr/engine/filter.h:65-74; CPU src/engine/filter.cpp:34-206,
 *      CUDA kernel_evaluate_predicates src/engine/filter_kernels.cu:22-64): AND of per-channel predicates on
 *      Float32 channels.  Produces a byte mask (1 = keep) instead of a compacted index list; the engine applies
 *      the mask inside its routing kernels, so a filtered ingest costs one extra byte per point. */
#define PCR_HIP_MAX_FILTER_PREDICATES 16
#define PCR_HIP_MAX_FILTER_SET 16          /* the reference's device limit (src/engine/filter_kernels.cu:15) */
enum {                                      /* pcr::CompareOp numbering (include/pcr/engine/filter.h:20-29) */
    PCR_HIP_CMP_EQUAL = 0, PCR_HIP_CMP_NOT_EQUAL = 1, PCR_HIP_CMP_LESS = 2, PCR_HIP_CMP_LESS_EQUAL = 3,
    PCR_HIP_CMP_GREATER = 4, PCR_HIP_CMP_GREATER_EQUAL = 5, PCR_HIP_CMP_IN_SET = 6, PCR_HIP_CMP_NOT_IN_SET = 7
};
typedef struct pcr_hip_predicate {
    const float* d_channel;
    int32_t op;
    float value;
    int32_t set_size;
    float set[PCR_HIP_MAX_FILTER_SET];
} pcr_hip_predicate;
/* d_mask[i] = all predicates hold for point i; *d_pass_count (optional, device u64, zeroed by the call) = survivors */
int pcr_hip_filter_mask(const pcr_hip_predicate* preds, int n_pred, uint64_t n, uint8_t* d_mask,
                        unsigned long long* d_pass_count, pcr_hip_stream s);
/* Points with d_mask[i] == 0 are ignored by every following scatter on this engine; NULL clears it. */
int pcr_hip_engine_set_point_mask(pcr_hip_engine* e, const uint8_t* d_mask);

/* ---- multi-GPU routing of an unpartitioned cloud.  No reference counterpart: the reference is single-device and
 *      its 1 B-point protocol feeds one pipeline (scripts/benchmarks/benchmark_billion_points.py:221-345).  With the
 *      grid row-block sharded over GPUs, a rank groups the points it was handed by OWNER (the part whose row range
 *      holds the point's centre row: GridConfig::world_to_cell, src/core/grid_config.cpp:24-43, exactly as the
 *      scatter kernels evaluate it), the groups travel point-to-point (RCCL all-to-all, pcr/distributed.py), and
 *      every rank ingests only points it owns.
 *   route_count:   d_dest[i] = owning part (0xFF: outside the grid / masked out / owned by nobody);
 *                  d_counts[p] (device u64, zeroed by the call) = points owned by part p.
 *                  row_splits: HOST array of nparts+1 ascending rows, part p owns [row_splits[p], row_splits[p+1]).
 *   route_scatter: regroups up to 8 arrays of 4- or 8-byte elements by owner; d_cursors[p] (device u64) holds the
 *                  first output slot of part p on entry (exclusive prefix sums of the counts) and is advanced.
 *                  Order inside a part is unspecified. */
#define PCR_HIP_MAX_ROUTE_PARTS 64
#define PCR_HIP_MAX_ROUTE_ARRAYS 8
int pcr_hip_route_count(const pcr_hip_grid* g, const int32_t* row_splits, int nparts,
                        const double* d_x, const double* d_y, const uint8_t* d_mask, uint64_t n,
                        uint8_t* d_dest, unsigned long long* d_counts, pcr_hip_stream s);
int pcr_hip_route_scatter(const uint8_t* d_dest, uint64_t n, int nparts, unsigned long long* d_cursors,
                          int narrays, const void* const* d_src, void* const* d_dst, const int32_t* elem_bytes,
                          pcr_hip_stream s);
/* max |d_values[i]| over the finite entries (0 for n == 0), synchronizes the stream.  Sizes the halo check of a
 * row-block shard whose Line glyph has a per-point half_length channel (its y reach is not capped by
 * max_radius_cells: glyph_kernels.cu:228-234). */
int pcr_hip_absmax_f32(const float* d_values, uint64_t n, float* h_result, pcr_hip_stream s);
/* The same over the points a filter mask keeps (d_mask may be NULL), with the 4-byte device word the reduction needs
 * supplied by the caller (NULL: allocated and freed inside, which synchronizes the whole device). */
int pcr_hip_absmax_f32_masked(const float* d_values, const uint8_t* d_mask, uint64_t n, uint32_t* d_scratch_word,
                              float* h_result, pcr_hip_stream s);
/* max(+v) and max(-v) over the finite entries the mask keeps (each >= 0; d_scratch_2words: 8 bytes of device memory).
 * The reference caps a Line's y half-extent with std::min(hy, cap), hy = half_length / cell_size_y (glyph_kernels.cu:228-234):
 * only the sign of half_length that makes hy POSITIVE is capped, so a shard's halo check needs the two sides apart. */
int pcr_hip_signed_max_f32_masked(const float* d_values, const uint8_t* d_mask, uint64_t n, uint32_t* d_scratch_2words,
                                  float* h_max_pos, float* h_max_neg, pcr_hip_stream s);

/* ---- multi-device: the exchange step of row-block shards, over RCCL / xGMI.  New work: the reference is single-device
 *      (cuda_device_id, include/pcr/engine/pipeline.h:68); SURVEY section 8b asks for these entry points, 8e fixes their
 *      shape.  One process per GPU; rank r owns rows [own_row0, own_row1) and keeps `halo` apron rows on each side inside
 *      its state window [state_row0, state_row0 + state_rows).
 *        halo_reduce        apron rows -> the neighbour that owns them (ncclSend/ncclRecv to rank +- 1 in one group),
 *                           merged there with the plane's op (add for SUM / WGT, fmaxf / fminf for MAX / MIN)
 *        allreduce_max_u32  the touched-tile flags (one word per reference tile)
 *      Both are enqueued on the caller's stream.  RCCL is loaded on first use (librccl.so.1); without it the calls
 *      return PCR_HIP_NOT_IMPLEMENTED and pcr_hip_comm_available() is 0.  Bootstrap: rank 0 makes a 128-byte id, the
 *      host carries it to the other ranks (MPI, a file, torch.distributed, ...), every rank calls _create. */
#define PCR_HIP_COMM_ID_BYTES 128
typedef struct pcr_hip_comm pcr_hip_comm;
typedef struct pcr_hip_halo_plane {
    float* d_plane;            /* state_rows x width floats */
    uint32_t kind;             /* PCR_HIP_PLANE_SUM / _WGT / _MAX / _MIN */
    uint32_t reserved_;
} pcr_hip_halo_plane;
int pcr_hip_comm_available(void);
int pcr_hip_comm_unique_id(uint8_t* id128);
int pcr_hip_comm_create(pcr_hip_comm** out, const uint8_t* id128, int rank, int world, int device);
int pcr_hip_comm_destroy(pcr_hip_comm* c);
int pcr_hip_comm_rank(const pcr_hip_comm* c, int* rank, int* world);
/* halo_reduce cannot hang on arguments that disagree between the ranks: every call first ALL-GATHERS what each rank brings
 * (one pcr_hip_halo_geom, posted whatever this rank's own arguments were; costs one small collective + a stream sync),
 * every rank judges the gathered records with the same pure function (pcr_hip_comm_halo_plan), and either all of them post
 * their sends and receives -- a receive sized from what its sender says it holds -- or all of them return
 * PCR_HIP_INVALID_ARGUMENT with the same message (a block shorter than a neighbour's apron, blocks that are not contiguous,
 * a rank that owns no rows, different widths / halos / planes, a rank with invalid arguments).  At most 8 planes per call. */
typedef struct pcr_hip_halo_geom {
    int32_t width, state_row0, state_rows, own_row0, own_row1, halo, nplanes;
    int32_t kinds;             /* plane kinds, 4 bits each, plane 0 in the low bits */
    int32_t valid;             /* 0: this rank's own arguments were unusable -- everyone refuses */
    int32_t reserved_;
} pcr_hip_halo_geom;
int pcr_hip_comm_halo_reduce(pcr_hip_comm* c, const pcr_hip_halo_plane* planes, int nplanes, int width,
                             int state_row0, int state_rows, int own_row0, int own_row1, int halo, pcr_hip_stream s);
/* The verdict on `world` gathered records, and rank's message sizes in rows (0: no message).  Pure host code: no
 * communicator, no device -- what the CPU tests drive. */
int pcr_hip_comm_halo_plan(const pcr_hip_halo_geom* all, int world, int rank, int* send_up_rows, int* send_dn_rows,
                           int* recv_up_rows, int* recv_dn_rows);
/* MAX over the ranks of one host integer (collective; synchronizes the stream): what a sharded ingest agrees on before
 * anything is accumulated, e.g. the Line reach of this round's clouds, so that every rank refuses together. */
int pcr_hip_comm_agree_max_i32(pcr_hip_comm* c, int32_t* h_inout, pcr_hip_stream s);
int pcr_hip_comm_allreduce_max_u32(pcr_hip_comm* c, uint32_t* d_words, int count, pcr_hip_stream s);
int pcr_hip_comm_allreduce_sum_f64(pcr_hip_comm* c, double* d_values, int count, pcr_hip_stream s);
int pcr_hip_comm_stats(const pcr_hip_comm* c, uint64_t* halo_reduces, uint64_t* bytes_sent);
/* Variable-size transfers (round 5), with the halo reduce's discipline: what every rank brings is all-gathered first (one
 * pcr_hip_xfer_geom each, posted whatever the rank's own arguments were), every rank passes the same verdict on the gathered
 * records (pcr_hip_comm_xfer_plan, pure host code), and either all post their sends and receives -- a receive sized from what
 * its sender announced -- or all return PCR_HIP_INVALID_ARGUMENT with the same message.  Up to 8 arrays travel in ONE group with
 * the same element counts (a cloud's x, y and channels; a result's bands); array a has elem_bytes[a]-byte elements (1..16).
 *   alltoallv       SURVEY section 8e "device-side partition + peer copy": d_send[a] holds this rank's elements grouped by
 *                   destination rank (h_send_counts[p] elements for rank p, in rank order: pcr_hip_route_scatter's output);
 *                   d_recv[a] receives the groups of all ranks in rank order; h_recv_counts[p] (optional) = elements from p.
 *                   recv_capacity = elements d_recv[a] can hold: a rank that would overflow makes every rank refuse.
 *   alltoall_counts the counts alone (every rank's h_send_counts row to every rank): sizes the receive buffers beforehand.
 *   gatherv         every rank's send_count elements per array to `root`, landing there in rank order (the row strips of a
 *                   sharded result -> one grid, src/engine/pipeline.cpp:1175-1186, 1351-1361 write ONE file).
 * A rank's own group is a device-to-device copy; a world of one needs no RCCL at all.  All on the caller's stream. */
#define PCR_HIP_MAX_XFER_ARRAYS 8
typedef struct pcr_hip_xfer_geom {
    int32_t narrays;
    int32_t elem_bytes[PCR_HIP_MAX_XFER_ARRAYS];
    int32_t root;              /* gatherv: the receiving rank; alltoallv: -1 */
    int32_t valid;             /* 0: this rank's own arguments were unusable -- everyone refuses */
    int32_t reserved_;
    uint64_t recv_capacity;    /* elements per array this rank can take */
    uint64_t send_counts[PCR_HIP_MAX_ROUTE_PARTS];   /* elements per array this rank sends to rank p */
} pcr_hip_xfer_geom;
int pcr_hip_comm_alltoall_counts(pcr_hip_comm* c, const uint64_t* h_send_counts, uint64_t* h_recv_counts, pcr_hip_stream s);
int pcr_hip_comm_alltoallv(pcr_hip_comm* c, int narrays, const void* const* d_send, void* const* d_recv, const int32_t* elem_bytes,
                           const uint64_t* h_send_counts, uint64_t recv_capacity, uint64_t* h_recv_counts, pcr_hip_stream s);
int pcr_hip_comm_gatherv(pcr_hip_comm* c, int narrays, const void* const* d_send, void* const* d_recv, const int32_t* elem_bytes,
                         uint64_t send_count, uint64_t recv_capacity, uint64_t* h_recv_counts, int root, pcr_hip_stream s);
/* send_offsets / recv_offsets: world + 1 entries (the last = totals); recv_counts: world entries. */
int pcr_hip_comm_xfer_plan(const pcr_hip_xfer_geom* all, int world, int rank, uint64_t* send_offsets, uint64_t* recv_counts,
                           uint64_t* recv_offsets);

/* Per-kernel timing with HIP events on the engine's stream (for roofline reporting).
 * While enabled every kernel the engine launches is bracketed by two events; _read drains the
 * events (synchronizes) and returns, per kernel name, launches and summed milliseconds. */
typedef struct pcr_hip_kernel_time {
    char name[48];
    uint32_t launches;
    double total_ms;
} pcr_hip_kernel_time;
int pcr_hip_engine_profile_enable(pcr_hip_engine* e, int on);
/* Every pair of events costs the stream ~4 us of serialisation (measured: 52 us on a 0.69 ms step with 7 kernels).
 * kernel_name != NULL / "": only launches timed under that name are bracketed (a roofline needs one kernel, and the
 * step around it should run as it does in production); NULL or "": all of them. */
int pcr_hip_engine_profile_only(pcr_hip_engine* e, const char* kernel_name);
int pcr_hip_engine_profile_read(pcr_hip_engine* e, pcr_hip_kernel_time* out, int capacity, int* count, int reset);

/* Point glyph: every valid point folds `value` into the planes named by plane_mask at its cell.
 * d_value may be NULL only when plane_mask == PCR_HIP_PLANE_WGT. */
int pcr_hip_scatter_point(pcr_hip_engine* e, uint32_t plane_mask, const pcr_hip_planes* planes,
                          const double* d_x, const double* d_y, const float* d_value, uint64_t n);
/* Line / Gaussian glyph: planes SUM (+= v*w) and/or WGT (+= w) only. */
int pcr_hip_scatter_glyph(pcr_hip_engine* e, const pcr_hip_glyph* glyph, uint32_t plane_mask,
                          const pcr_hip_planes* planes,
                          const double* d_x, const double* d_y, const float* d_value, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
