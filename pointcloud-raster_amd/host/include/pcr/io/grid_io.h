// pcr/io/grid_io.h -- GeoTIFF output of finalized grids (drop-in for the reference's
// include/pcr/io/grid_io.h:15-88: same names, options and argument meaning).
//
// The reference writes through GDAL (src/io/grid_io.cpp), which this image does not have; here the
// files are written directly: little-endian TIFF 6.0 or BigTIFF, IEEE float32 samples, one plane per
// band (PlanarConfiguration = 2, the layout of Grid), strips or tiles, compression NONE / LZW /
// DEFLATE (zlib), GeoTIFF georeferencing (ModelPixelScale + ModelTiepoint, or ModelTransformation
// for south-up grids; GeoKeyDirectory with the EPSG code when the CRS has one) and GDAL's own
// GDAL_METADATA (band descriptions) and GDAL_NODATA ("nan") tags, so that GDAL-based readers see what
// the reference's files show: band names, NaN nodata, geotransform, EPSG.
// Not supported: ZSTD, cloud_optimized (overviews) -- NotImplemented.
#pragma once

#include "pcr/core/grid_config.h"
#include "pcr/core/types.h"

#include <memory>
#include <string>
#include <vector>

namespace pcr {

class Grid;

struct GeoTiffOptions {
    bool cloud_optimized = false;
    std::string compress = "LZW";          // NONE, LZW, DEFLATE
    int compress_level = 6;                // DEFLATE
    int tile_width = 256;                  // internal TIFF tile (multiple of 16); 0 = strips
    int tile_height = 256;
    bool bigtiff = true;
    std::string overview_resampling = "average";
};

/// Grid must be host-resident and match `config` (width x height).  Band names -> band descriptions.
Status write_geotiff(const std::string& path, const Grid& grid, const GridConfig& config,
                     const GeoTiffOptions& options = {});

/// Incremental assembly: reference tiles (GridConfig tiling) in any order; tiles never written
/// read back as nodata.
class TiledGeoTiffWriter {
public:
    ~TiledGeoTiffWriter();
    static std::unique_ptr<TiledGeoTiffWriter> open(const std::string& path, const GridConfig& config,
                                                    const std::vector<std::string>& band_names,
                                                    const GeoTiffOptions& options = {});
    /// `data`: host, band-sequential [band][row][col] over the tile's own cell range.
    Status write_tile(TileIndex tile, const float* data, int num_bands);
    Status close();

private:
    TiledGeoTiffWriter() = default;
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

/// Reads files written by this writer (and plain float32 strip/tile TIFFs with the same
/// compressions): size, band count, CRS (EPSG only) and bounds from the georeferencing tags.
Status read_geotiff_info(const std::string& path, int& width, int& height, int& num_bands, CRS& crs, BBox& bounds);
Status read_geotiff_band(const std::string& path, int band_index, float* data, int width, int height);
/// Extension: band descriptions stored by the writer (empty strings when absent).
Status read_geotiff_band_names(const std::string& path, std::vector<std::string>& names);

}  // namespace pcr
