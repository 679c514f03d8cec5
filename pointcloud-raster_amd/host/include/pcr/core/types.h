// pcr/core/types.h -- value types of the public API (drop-in for the reference's
// include/pcr/core/types.h: same names, enumerator order, fields and defaults).
// Device helpers are backed by the HIP C-ABI (include/pcr_hip.h); the `cuda_*` spellings
// and StatusCode::CudaError are kept so existing callers compile unchanged.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <string>

namespace pcr {

enum class DataType : uint8_t { Float32, Float64, Int32, UInt32, Int16, UInt16, UInt8 };

size_t data_type_size(DataType dt);

// Only Sum, Max, Min, Average, WeightedAverage and Count are executable (as in the
// reference, src/ops/reduction_registry.cpp:173-184); the rest are declared for API parity.
enum class ReductionType : uint8_t {
    Sum, Max, Min, Average, WeightedAverage, Count,
    Median, Percentile, MostRecent, PriorityMerge, Custom
};

enum class MemoryLocation : uint8_t { Host, HostPinned, Device };

enum class StatusCode : uint8_t {
    Ok, InvalidArgument, OutOfMemory, CudaError, IoError, CrsError, NotImplemented
};

struct Status {
    StatusCode code = StatusCode::Ok;
    std::string message;

    bool ok() const { return code == StatusCode::Ok; }
    static Status success() { return {}; }
    static Status error(StatusCode c, const std::string& msg) { return {c, msg}; }
};

struct BBox {
    double min_x = std::numeric_limits<double>::max();
    double min_y = std::numeric_limits<double>::max();
    double max_x = std::numeric_limits<double>::lowest();
    double max_y = std::numeric_limits<double>::lowest();

    void expand(double x, double y);
    void expand(const BBox& other);
    bool contains(double x, double y) const;      // inclusive on all four edges

    double width() const { return max_x - min_x; }
    double height() const { return max_y - min_y; }
    bool valid() const { return max_x >= min_x && max_y >= min_y; }
};

// CRS is carried, never interpreted, on the ingest->finalize path.  Without PROJ in the
// build, projected/geographic are answered from the WKT keywords / EPSG ranges only.
struct CRS {
    std::string wkt;
    int epsg = 0;

    bool is_projected() const;
    bool is_geographic() const;
    bool is_valid() const { return !wkt.empty() || epsg != 0; }

    static CRS from_epsg(int code);
    static CRS from_wkt(const std::string& wkt_str);
    bool equivalent_to(const CRS& other) const;
};

struct NoDataPolicy {
    float value = std::nanf("");
    bool use_nan = true;
    float sentinel() const { return use_nan ? std::nanf("") : value; }
};

struct TileIndex {
    int row = 0;
    int col = 0;
    bool operator==(const TileIndex& o) const { return row == o.row && col == o.col; }
    bool operator<(const TileIndex& o) const { return row < o.row || (row == o.row && col < o.col); }
};

// Device discovery (reference names; answered by the HIP runtime through the C-ABI).
bool cuda_is_compiled();                 // true: this build always carries the HIP engine
bool cuda_device_available();
int cuda_device_count();
std::string cuda_device_name(int device_id = 0);
bool cuda_get_memory_info(size_t* free_bytes, size_t* total_bytes, int device_id = 0);

}  // namespace pcr
