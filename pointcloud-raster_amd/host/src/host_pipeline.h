// host_pipeline.h -- (internal) Pipeline::Host: the pipeline on the host engine (host_engine.h), behind ExecutionMode::CPU,
// Auto without a GPU and gpu_fallback_to_cpu -- what the reference does in those cases (src/engine/pipeline.cpp:100-131).
// Same validation, messages, band naming, progress callback and `.pcrt` checkpoints as the HIP pipeline.
#pragma once

#include "host_engine.h"
#include "pcr/core/grid.h"
#include "pcr/engine/pipeline.h"
#include "pipeline_common.h"

#include <chrono>
#include <memory>
#include <string>
#include <vector>

namespace pcr {

struct Pipeline::Host {
    struct Group {                       // one pass over the points: same grouping as the HIP pipeline
        std::string value_channel;
        GlyphSpec glyph;
        detail::HostPlanes planes;
        uint32_t mask = 0;
    };
    struct Output {
        int group = 0;
        ReductionType type = ReductionType::Sum;
        std::string band_name;
    };

    PipelineConfig cfg;
    std::unique_ptr<detail::HostEngine> engine;
    std::vector<Group> groups;
    std::vector<Output> outputs;
    std::unique_ptr<Grid> result;
    bool finalized = false;
    ProgressCallback callback;
    size_t collections = 0, points = 0;
    ScatterInfo last{};
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();

    Status init();
    Status ingest(const PointCloud& cloud);
    Status finalize();
    Status save_state(const std::string& dir);
    Status load_state(const std::string& dir);
    ProgressInfo stats() const;
    std::vector<detail::StateOutput> state_outputs() const;
};

}  // namespace pcr
