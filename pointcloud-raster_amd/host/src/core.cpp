// core.cpp -- value types: DataType sizes, BBox, CRS, GridConfig, device discovery helpers.
// Behavioural contract: the reference's src/core/types.cpp:13-43 and
// src/core/grid_config.cpp:7-147 (inclusive bounds, floor-of-division cell lookup with
// clamping, edge-clamped tiles).  CRS handling carries EPSG/WKT without PROJ.
#include "pcr/core/grid_config.h"
#include "pcr/core/types.h"

#include "pcr_hip.h"

#include <algorithm>
#include <cmath>

namespace pcr {

size_t data_type_size(DataType dt) {
    switch (dt) {
        case DataType::Float64: return 8;
        case DataType::Float32: case DataType::Int32: case DataType::UInt32: return 4;
        case DataType::Int16: case DataType::UInt16: return 2;
        case DataType::UInt8: return 1;
    }
    return 0;
}

// ---- BBox ---------------------------------------------------------------------------
void BBox::expand(double x, double y) {
    if (x < min_x) min_x = x;
    if (y < min_y) min_y = y;
    if (x > max_x) max_x = x;
    if (y > max_y) max_y = y;
}

void BBox::expand(const BBox& other) {
    if (!other.valid()) return;
    expand(other.min_x, other.min_y);
    expand(other.max_x, other.max_y);
}

bool BBox::contains(double x, double y) const {
    return x >= min_x && x <= max_x && y >= min_y && y <= max_y;
}

// ---- CRS (no PROJ: keyword / EPSG-range heuristics) -----------------------------------
static bool wkt_has(const std::string& wkt, const char* key) { return wkt.find(key) != std::string::npos; }

bool CRS::is_projected() const {
    if (wkt_has(wkt, "PROJCS") || wkt_has(wkt, "PROJCRS")) return true;
    if (!wkt.empty()) return false;
    // EPSG projected ranges most callers use: UTM (326xx/327xx), Web Mercator, state planes...
    return epsg == 3857 || (epsg >= 2000 && epsg < 4000) || (epsg >= 20000 && epsg < 33000);
}

bool CRS::is_geographic() const {
    if (wkt_has(wkt, "PROJCS") || wkt_has(wkt, "PROJCRS")) return false;
    if (wkt_has(wkt, "GEOGCS") || wkt_has(wkt, "GEOGCRS") || wkt_has(wkt, "GEODCRS")) return true;
    return epsg == 4326 || epsg == 4269 || epsg == 4979 || (epsg >= 4000 && epsg < 5000 && !is_projected());
}

CRS CRS::from_epsg(int code) {
    CRS c;
    c.epsg = code;
    return c;
}

CRS CRS::from_wkt(const std::string& wkt_str) {
    CRS c;
    c.wkt = wkt_str;
    return c;
}

bool CRS::equivalent_to(const CRS& other) const {
    if (epsg != 0 && other.epsg != 0) return epsg == other.epsg;
    if (!wkt.empty() && !other.wkt.empty()) return wkt == other.wkt;
    return false;
}

// ---- device discovery ---------------------------------------------------------------------
bool cuda_is_compiled() { return true; }

int cuda_device_count() {
    int n = 0;
    if (pcr_hip_device_count(&n) != PCR_HIP_OK) return 0;
    return n;
}

bool cuda_device_available() { return cuda_device_count() > 0; }

std::string cuda_device_name(int device_id) {
    char buf[256];
    if (pcr_hip_device_name(device_id, buf, sizeof buf) != PCR_HIP_OK) return "Unknown GPU";
    return buf;
}

bool cuda_get_memory_info(size_t* free_bytes, size_t* total_bytes, int device_id) {
    if (pcr_hip_set_device(device_id) != PCR_HIP_OK) return false;
    return pcr_hip_mem_info(free_bytes, total_bytes) == PCR_HIP_OK;
}

// ---- GridConfig ---------------------------------------------------------------------------
void GridConfig::compute_dimensions() {
    if (!bounds.valid()) {
        width = height = tiles_x = tiles_y = 0;
        return;
    }
    width = static_cast<int>(std::ceil(bounds.width() / std::fabs(cell_size_x)));
    height = static_cast<int>(std::ceil(bounds.height() / std::fabs(cell_size_y)));
    tiles_x = (width + tile_width - 1) / tile_width;
    tiles_y = (height + tile_height - 1) / tile_height;
}

bool GridConfig::world_to_cell(double wx, double wy, int& col, int& row) const {
    if (!bounds.contains(wx, wy)) return false;
    int c = static_cast<int>(std::floor((wx - bounds.min_x) / cell_size_x));
    int r = static_cast<int>(std::floor((wy - bounds.max_y) / cell_size_y));
    col = std::clamp(c, 0, std::max(width - 1, 0));
    row = std::clamp(r, 0, std::max(height - 1, 0));
    return true;
}

void GridConfig::cell_to_world(int col, int row, double& wx, double& wy) const {
    wx = bounds.min_x + (col + 0.5) * cell_size_x;
    wy = bounds.max_y + (row + 0.5) * cell_size_y;
}

TileIndex GridConfig::cell_to_tile(int col, int row) const {
    TileIndex t;
    t.row = row / tile_height;
    t.col = col / tile_width;
    return t;
}

void GridConfig::tile_cell_range(TileIndex idx, int& col_start, int& row_start,
                                 int& col_count, int& row_count) const {
    col_start = idx.col * tile_width;
    row_start = idx.row * tile_height;
    col_count = std::min(tile_width, width - col_start);
    row_count = std::min(tile_height, height - row_start);
}

BBox GridConfig::tile_bounds(TileIndex idx) const {
    int c0, r0, nc, nr;
    tile_cell_range(idx, c0, r0, nc, nr);
    BBox b;
    b.min_x = bounds.min_x + c0 * cell_size_x;
    b.max_x = bounds.min_x + (c0 + nc) * cell_size_x;
    b.max_y = bounds.max_y + r0 * cell_size_y;
    b.min_y = bounds.max_y + (r0 + nr) * cell_size_y;
    return b;
}

void GridConfig::gdal_geotransform(double gt[6]) const {
    gt[0] = bounds.min_x;
    gt[1] = cell_size_x;
    gt[2] = 0.0;
    gt[3] = bounds.max_y;
    gt[4] = 0.0;
    gt[5] = cell_size_y;
}

Status GridConfig::validate() const {
    if (!bounds.valid()) return Status::error(StatusCode::InvalidArgument, "Invalid bounds: max < min");
    if (cell_size_x == 0.0 || cell_size_y == 0.0)
        return Status::error(StatusCode::InvalidArgument, "Cell size cannot be zero");
    if (tile_width <= 0 || tile_height <= 0)
        return Status::error(StatusCode::InvalidArgument, "Tile dimensions must be positive");
    if (width <= 0 || height <= 0)
        return Status::error(StatusCode::InvalidArgument,
                             "Grid dimensions not computed or invalid. Call compute_dimensions()");
    if (!crs.is_valid()) return Status::error(StatusCode::CrsError, "CRS is not valid");
    return Status::success();
}

}  // namespace pcr
