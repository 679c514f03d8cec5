// host_engine.h -- (internal) the engine behind ExecutionMode::CPU: the same ingest -> finalize contract as the HIP engine
// on the host's cores.
//
// The reference's CPU path (src/engine/pipeline.cpp:283-770) routes, SORTS the cloud by (tile, cell) with a serial
// std::sort, cuts it into per-tile batches and folds every point into the tile state under `omp critical`
// (src/ops/reduction_registry.cpp:63-92) or `omp atomic` (glyphs, src/engine/glyph_kernels.cu:36-74): its thread scaling is
// flat to negative and its sums depend on the thread schedule.  This engine keeps the reference's ARITHMETIC cell for cell --
// world_to_cell with inclusive bounds and clamp (src/core/grid_config.cpp:24-43), the ops of include/pcr/ops/builtin_ops.h,
// the Gaussian and Bresenham footprints of glyph_kernels.cu:79-281 clipped to the reference tile of the centre cell -- on a
// different machine: no sort, no batches, no atomics.  The grid's rows are cut into STRIPES, every stripe belongs to one task,
// a point (or a footprint's rows) is listed with every stripe it reaches, and a stripe folds its list in point order into the
// planes it owns.  Deterministic whatever the thread count: a cell always sees its contributions in ascending point index.
//
// Not part of oracle/ and shares no file with it: the oracle checks this engine like it checks the HIP one
// (tests/test_host_engine.py).
#pragma once

#include "pcr/core/grid_config.h"
#include "pcr/core/types.h"
#include "pcr/engine/glyph.h"

#include <cstddef>
#include <cstdint>
#include <vector>

namespace pcr {
namespace detail {

// plane order as on the device: 0 = sum (of v * w), 1 = weight (of w), 2 = max, 3 = min; bit p of `mask` = plane p exists
struct HostPlanes {
    uint32_t mask = 0;
    std::vector<float> plane[4];
};

struct HostGlyphArrays {           // per-point glyph channels (null: the GlyphSpec default)
    const float* direction = nullptr;
    const float* half_length = nullptr;
    const float* sigma_x = nullptr;
    const float* sigma_y = nullptr;
    const float* rotation = nullptr;
};

class HostEngine {
public:
    HostEngine(const GridConfig& grid, int threads);

    int threads() const { return threads_; }
    /// Identity into the planes `mask` names (0 / 0 / -FLT_MAX / FLT_MAX), whole grid.
    void init_planes(HostPlanes& p, uint32_t mask) const;

    /// Routes one cloud (at most 2^31 points): cell of every point, touched tiles.  keep[i] == 0: point i does not exist
    /// (the filter stage); keep may be null.  Returns the points inside the grid.  The scatter calls below fold THIS cloud.
    size_t route(const double* x, const double* y, const uint8_t* keep, size_t n);
    void scatter_point(HostPlanes& p, const float* v);
    /// Planes 0 / 1 only (glyph reductions are Sum, Count, Average, WeightedAverage).
    void scatter_glyph(HostPlanes& p, const GlyphSpec& glyph, const HostGlyphArrays& arr, const float* v);

    /// Op::finalize per cell where the reference tile is touched, NaN elsewhere (src/engine/pipeline.cpp:1204-1222).
    void finalize(const HostPlanes& p, ReductionType type, float* band) const;

    std::vector<uint32_t>& touched() { return touched_; }          // tiles_x * tiles_y flags
    const std::vector<uint32_t>& touched() const { return touched_; }
    size_t tiles_active() const;

private:
    struct Stripes;                       // per-stripe point lists of the cloud being folded
    template <class RowsOf> void build_lists(std::vector<uint32_t>& list, std::vector<size_t>& first, RowsOf rows_of) const;

    GridConfig g_;
    int threads_ = 1;
    int W_ = 0, H_ = 0, tiles_x_ = 0, tiles_y_ = 0;
    int stripe_rows_ = 1, nstripes_ = 1;
    // the routed cloud
    const double* x_ = nullptr;
    const double* y_ = nullptr;
    size_t n_ = 0;
    std::vector<int32_t> col_, row_;      // row < 0: the point is outside the grid (or filtered out)
    std::vector<uint32_t> touched_;
};

}  // namespace detail
}  // namespace pcr
