// pipeline_common.h -- (internal) what the engines behind Pipeline share: which reductions exist, which planes they
// need, how ReductionSpecs are grouped into passes over the points, and the `.pcrt` tile files of a state window.
#pragma once

#include "pcr/core/grid_config.h"
#include "pcr/engine/pipeline.h"

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace pcr {
namespace detail {

// plane bits = PCR_HIP_PLANE_* of include/pcr_hip.h (sum, weight, max, min)
constexpr uint32_t kPlaneBits[4] = {1u, 2u, 4u, 8u};

// Only these six have an implementation in the reference (src/ops/reduction_registry.cpp:173-184); Pipeline::create
// refuses the others (src/engine/pipeline.cpp:229-233).
inline bool registered(ReductionType t) {
    switch (t) {
        case ReductionType::Sum: case ReductionType::Max: case ReductionType::Min:
        case ReductionType::Average: case ReductionType::WeightedAverage: case ReductionType::Count:
            return true;
        default:
            return false;
    }
}

// src/engine/pipeline.cpp:500-508
inline bool glyph_reduction_ok(ReductionType t) {
    return t == ReductionType::WeightedAverage || t == ReductionType::Average ||
           t == ReductionType::Sum || t == ReductionType::Count;
}

inline uint32_t planes_for(ReductionType t) {
    switch (t) {
        case ReductionType::Sum: return kPlaneBits[0];
        case ReductionType::Count: return kPlaneBits[1];
        case ReductionType::Max: return kPlaneBits[2];
        case ReductionType::Min: return kPlaneBits[3];
        default: return kPlaneBits[0] | kPlaneBits[1];      // Average, WeightedAverage
    }
}

inline bool same_glyph(const GlyphSpec& a, const GlyphSpec& b) {
    if (a.type != b.type) return false;
    if (a.type == GlyphType::Point) return true;
    return a.direction_channel == b.direction_channel && a.default_direction == b.default_direction &&
           a.half_length_channel == b.half_length_channel && a.default_half_length == b.default_half_length &&
           a.sigma_x_channel == b.sigma_x_channel && a.default_sigma_x == b.default_sigma_x &&
           a.sigma_y_channel == b.sigma_y_channel && a.default_sigma_y == b.default_sigma_y &&
           a.rotation_channel == b.rotation_channel && a.default_rotation == b.default_rotation &&
           a.max_radius_cells == b.max_radius_cells;
}

// planes a reduction's reference state is made of, in the reference's field order
// (builtin_ops.h: Sum{sum}, Max{val}, Min{val}, Count{count}, Average{sum,count}, WeightedAverage{wsum,wgt})
inline int state_planes_of(ReductionType t, int out[2]) {
    switch (t) {
        case ReductionType::Sum: out[0] = 0; return 1;
        case ReductionType::Count: out[0] = 1; return 1;
        case ReductionType::Max: out[0] = 2; return 1;
        case ReductionType::Min: out[0] = 3; return 1;
        default: out[0] = 0; out[1] = 1; return 2;
    }
}

inline std::string default_band_name(const ReductionSpec& r) {
    return r.output_band_name.empty() ? r.value_channel + "_" + std::to_string(static_cast<int>(r.type)) : r.output_band_name;
}

// One ReductionSpec as the checkpoint code sees it.
struct StateOutput {
    int group;
    ReductionType type;
};

// ReductionSpecs -> accumulation groups (same value channel through the same glyph = one pass, one set of planes).
struct Grouping {
    std::vector<uint32_t> masks;             // per group: planes it keeps
    std::vector<StateOutput> outputs;        // per ReductionSpec
};
inline Grouping group_reductions(const std::vector<ReductionSpec>& reductions) {
    Grouping out;
    std::vector<const ReductionSpec*> first;
    for (const auto& r : reductions) {
        int gi = -1;
        for (size_t k = 0; k < first.size(); ++k)
            if (first[k]->value_channel == r.value_channel && same_glyph(first[k]->glyph, r.glyph)) gi = (int)k;
        if (gi < 0) { first.push_back(&r); out.masks.push_back(0u); gi = (int)first.size() - 1; }
        out.masks[(size_t)gi] |= planes_for(r.type);
        out.outputs.push_back({gi, r.type});
    }
    return out;
}

// Host copies of state planes over the row WINDOW [row0, row0 + rows) of the grid (whole tile rows for everything below):
// plane(group, p) -> rows x width floats, or null when the group has no plane p.
struct StateWindow {
    int row0 = 0, rows = 0;
    int own_row0 = -1, own_row1 = -1;      // >= 0: only tiles whose rows lie inside [own_row0, own_row1) (a shard with apron rows)
    std::function<float*(int group, int p)> plane;
};

/// "<dir>" for a single reduction (the reference's layout), "<dir>/reduction_<r>" otherwise.
std::string reduction_state_dir(const std::string& dir, size_t r, size_t n_outputs);

// What every engine checks of a cloud before it touches any state, with the reference's messages: the filter's channels
// (filter_points, src/engine/filter.cpp:101-123), every reduction's value channel and glyph / reduction pairing
// (src/engine/pipeline.cpp:365-378, 500-508).  max_set / max_predicates: the device filter's limits (0 = none, host engine).
Status validate_cloud(const PipelineConfig& cfg, const PointCloud& cloud, size_t max_set, size_t max_predicates);
/// One `.pcrt` file per touched reference tile inside the window and per output (src/io/tile_state_io.cpp:45-95;
/// file name src/io/tile_state_io.cpp:197-211).  touched: tiles_x * tiles_y flags of the whole grid.
Status write_state_tiles(const GridConfig& g, const std::vector<StateOutput>& outputs, const StateWindow& w,
                         const std::vector<uint32_t>& touched, const std::string& dir, std::vector<std::string>* written = nullptr);
/// The reverse: every matching file inside the window is copied into the planes and its tile marked touched; files that do
/// not describe their tile of their reduction are ignored like the reference's tile manager ignores them
/// (src/engine/tile_manager.cpp:272-320).  *loaded = files taken.
Status read_state_tiles(const GridConfig& g, const std::vector<StateOutput>& outputs, const StateWindow& w,
                        std::vector<uint32_t>& touched, const std::string& dir, size_t* loaded);

}  // namespace detail
}  // namespace pcr
