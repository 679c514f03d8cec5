// pcr/io/point_cloud_io.h -- point-cloud files either side of Pipeline::ingest (drop-in for the
// reference's include/pcr/io/point_cloud_io.h:14-99: same names, arguments and error behaviour).
//
// Native PCRP format (reference header, point_cloud_io.h:22-39), little-endian, packed:
//     u32 magic "PCRP" | u32 version = 1 | u64 num_points | u32 num_channels |
//     u32 crs_wkt_len | char crs_wkt[crs_wkt_len] |
//     { u16 name_len | char name[name_len] | u8 dtype (DataType) } x num_channels
//   body, SoA:  f64 x[num_points] | f64 y[num_points] | <dtype> channel[num_points] x num_channels
// CSV: header row "x,y,<channel>,...", values printed with 15 significant digits; every extra
// column is read back as Float64 (point_cloud_io.cpp:286-461).  LAS/LAZ: NotImplemented, as upstream.
//
// Extensions (MI355X build): read_point_cloud can deliver the cloud in page-locked host memory or
// straight in HBM (`location`), and the streaming reader returns the right rows for every chunk --
// the reference's PCRP chunk reader advances through the SoA body as if it were AoS
// (point_cloud_io.cpp:575-612) and only returns valid data when one chunk covers the whole file.
#pragma once

#include "pcr/core/point_cloud.h"
#include "pcr/core/types.h"

#include <memory>
#include <string>
#include <vector>

namespace pcr {

enum class PointCloudFormat : uint8_t { PCR_Binary, CSV, LAS, LAZ, Auto };

struct PointCloudInfo {
    size_t num_points = 0;
    std::vector<ChannelDesc> channels;
    CRS crs;
    BBox bounds;                      // empty: neither format stores it
};

/// Whole file -> PointCloud (nullptr on any failure, as upstream).  `location`: Host (default),
/// HostPinned (file read directly into page-locked memory) or Device (pinned staging, one H2D copy).
std::unique_ptr<PointCloud> read_point_cloud(const std::string& path,
                                             PointCloudFormat format = PointCloudFormat::Auto,
                                             MemoryLocation location = MemoryLocation::Host);

Status read_point_cloud_info(const std::string& path, PointCloudInfo& info,
                             PointCloudFormat format = PointCloudFormat::Auto);

/// Cloud must be host-resident (Host or HostPinned).
Status write_point_cloud(const std::string& path, const PointCloud& cloud,
                         PointCloudFormat format = PointCloudFormat::PCR_Binary);

class PointCloudReader {
public:
    ~PointCloudReader();
    static std::unique_ptr<PointCloudReader> open(const std::string& path,
                                                  PointCloudFormat format = PointCloudFormat::Auto);
    const PointCloudInfo& info() const;
    /// Next chunk of up to `max_points` into `cloud` (host-resident, capacity >= max_points; channels
    /// are added on first use).  Returns the number of points read, 0 at end of file.
    size_t read_chunk(PointCloud& cloud, size_t max_points);
    Status rewind();
    bool eof() const;

private:
    PointCloudReader() = default;
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace pcr
