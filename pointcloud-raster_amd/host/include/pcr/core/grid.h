// pcr/core/grid.h -- multi-band row-major raster (drop-in for the reference's
// include/pcr/core/grid.h).  Unlike the reference, Device grids are implemented: the
// pipeline can leave its finalized bands in HBM (PipelineConfig::result_location).
#pragma once

#include "pcr/core/grid_config.h"
#include "pcr/core/types.h"

#include <memory>
#include <string>
#include <vector>

namespace pcr {

struct BandDesc {
    std::string name;
    DataType dtype = DataType::Float32;
    bool is_state = false;
};

class Grid {
public:
    Grid() = default;
    ~Grid();

    static std::unique_ptr<Grid> create(int cols, int rows, const std::vector<BandDesc>& bands,
                                        MemoryLocation loc = MemoryLocation::Host);
    static std::unique_ptr<Grid> create_for_tile(const GridConfig& config, TileIndex tile,
                                                 const std::vector<BandDesc>& bands,
                                                 MemoryLocation loc = MemoryLocation::Host);

    /// Extension: a Host grid (location() == Host, numpy-viewable) whose bands are page-locked,
    /// so device-to-host copies of finalized bands run at full link speed.
    static std::unique_ptr<Grid> create_host_page_locked(int cols, int rows, const std::vector<BandDesc>& bands);

    int num_bands() const;
    BandDesc band_desc(int band_index) const;
    int band_index(const std::string& name) const;

    void* band_data(int band_index);
    const void* band_data(int band_index) const;
    float* band_f32(int band_index);
    const float* band_f32(int band_index) const;
    float* band_f32(const std::string& name);
    const float* band_f32(const std::string& name) const;

    int cols() const;
    int rows() const;
    int64_t cell_count() const;
    MemoryLocation location() const;

    Status fill(float value);
    Status fill_band(int band_index, float value);

    std::unique_ptr<Grid> to(MemoryLocation dst) const;
    std::unique_ptr<Grid> to_device_async(void* stream) const;
    Status copy_from(const Grid& other, void* stream = nullptr);

    std::vector<uint8_t> valid_mask(int band_index = 0) const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace pcr
