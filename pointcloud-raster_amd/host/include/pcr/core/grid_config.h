// pcr/core/grid_config.h -- the output raster's geometry (drop-in for the reference's
// include/pcr/core/grid_config.h).  world = origin + cell * cell_size, origin at
// (bounds.min_x, bounds.max_y); cell_size_y is negative on north-up grids.
#pragma once

#include "pcr/core/types.h"

namespace pcr {

struct GridConfig {
    BBox bounds;
    CRS crs;
    double cell_size_x = 1.0;
    double cell_size_y = -1.0;
    int width = 0;              // columns, set by compute_dimensions()
    int height = 0;             // rows
    NoDataPolicy nodata;
    int tile_width = 4096;      // reference tiling: clip rectangle of glyphs + "touched" granularity
    int tile_height = 4096;
    int tiles_x = 0;
    int tiles_y = 0;

    void compute_dimensions();
    bool world_to_cell(double wx, double wy, int& col, int& row) const;
    void cell_to_world(int col, int row, double& wx, double& wy) const;    // cell centre
    TileIndex cell_to_tile(int col, int row) const;
    BBox tile_bounds(TileIndex idx) const;
    void tile_cell_range(TileIndex idx, int& col_start, int& row_start,
                         int& col_count, int& row_count) const;
    int total_tiles() const { return tiles_x * tiles_y; }
    int64_t total_cells() const { return static_cast<int64_t>(width) * height; }
    void gdal_geotransform(double gt[6]) const;
    Status validate() const;
};

}  // namespace pcr
