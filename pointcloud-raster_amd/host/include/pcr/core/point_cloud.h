// pcr/core/point_cloud.h -- SoA point cloud: f64 x/y + named channels, on Host, pinned
// host or Device memory (drop-in for the reference's include/pcr/core/point_cloud.h).
// Device and pinned storage come from the HIP C-ABI.
#pragma once

#include "pcr/core/types.h"

#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace pcr {

struct ChannelDesc {
    std::string name;
    DataType dtype = DataType::Float32;
    size_t offset = 0;
};

class PointCloud {
public:
    PointCloud() = default;
    ~PointCloud();

    static std::unique_ptr<PointCloud> create(size_t capacity, MemoryLocation loc = MemoryLocation::Host);
    static std::unique_ptr<PointCloud> wrap(double* x, double* y, size_t count,
                                            MemoryLocation loc = MemoryLocation::Host);

    Status add_channel(const std::string& name, DataType dtype = DataType::Float32);
    bool has_channel(const std::string& name) const;
    const ChannelDesc* channel(const std::string& name) const;
    std::vector<std::string> channel_names() const;

    double* x();
    const double* x() const;
    double* y();
    const double* y() const;
    void* channel_data(const std::string& name);
    const void* channel_data(const std::string& name) const;
    float* channel_f32(const std::string& name);
    const float* channel_f32(const std::string& name) const;
    int32_t* channel_i32(const std::string& name);
    const int32_t* channel_i32(const std::string& name) const;

    size_t count() const;
    size_t capacity() const;
    MemoryLocation location() const;
    CRS crs() const;
    void set_crs(const CRS& crs);
    Status resize(size_t new_count);

    std::unique_ptr<PointCloud> to(MemoryLocation dst) const;
    std::unique_ptr<PointCloud> to_device_async(void* stream) const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace pcr
