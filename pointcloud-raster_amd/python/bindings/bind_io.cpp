// bind_io.cpp -- names of the reference's I/O layer (GeoTIFF via GDAL, PCRP/CSV/LAS readers),
// kept so that `import pcr` exposes the same symbols.  I/O is outside the accelerated
// ingest->finalize path of this build: every call raises RuntimeError.
#include "common.h"

#include "pcr/core/grid.h"
#include "pcr/core/grid_config.h"
#include "pcr/core/point_cloud.h"
#include "pcr/io/tile_state_io.h"

#include <vector>

using namespace pcr;

namespace {

enum class PointCloudFormat : uint8_t { PCR_Binary, CSV, LAS, LAZ, Auto };

struct GeoTiffOptions {
    bool cloud_optimized = false;
    std::string compress = "LZW";
    int compress_level = 6;
    int tile_width = 256;
    int tile_height = 256;
    bool bigtiff = true;
    std::string overview_resampling = "average";
};

struct PointCloudInfo {
    size_t num_points = 0;
    std::vector<ChannelDesc> channels;
    CRS crs;
    BBox bounds;
};

struct PointCloudReader {};

[[noreturn]] void unavailable(const char* what) {
    throw std::runtime_error(std::string(what) + ": file I/O is not part of this build "
                             "(only Pipeline.ingest/finalize is accelerated; use numpy/rasterio for files)");
}

}  // namespace

void bind_io(py::module_& m) {
    // `.pcrt` tile-state files (the reference's checkpoint format) -- implemented, unlike the rest of I/O
    m.def("write_tile_state", [](const std::string& path, int tile_row, int tile_col,
                                 py::array_t<float, py::array::c_style | py::array::forcecast> state,
                                 ReductionType type) {
        auto b = state.request();
        if (b.ndim != 3) throw std::runtime_error("write_tile_state: state must be [state_floats, rows, cols]");
        TileIndex t;
        t.row = tile_row;
        t.col = tile_col;
        raise_if_error(write_tile_state(path, t, (int)b.shape[2], (int)b.shape[1], (int)b.shape[0], type,
                                        static_cast<const float*>(b.ptr)));
    }, py::arg("path"), py::arg("tile_row"), py::arg("tile_col"), py::arg("state"), py::arg("type"));
    m.def("read_tile_state", [](const std::string& path) {
        TileIndex t;
        int cols = 0, rows = 0, k = 0;
        ReductionType type;
        raise_if_error(read_tile_state_header(path, t, cols, rows, k, type));
        py::array_t<float> out({k, rows, cols});
        raise_if_error(read_tile_state(path, t, cols, rows, k, type, out.mutable_data()));
        return py::make_tuple(t.row, t.col, out, type);
    }, py::arg("path"));
    m.def("tile_state_filename", [](const std::string& dir, int tile_row, int tile_col) {
        TileIndex t;
        t.row = tile_row;
        t.col = tile_col;
        return tile_state_filename(dir, t);
    });

    py::enum_<PointCloudFormat>(m, "PointCloudFormat")
        .value("PCR_Binary", PointCloudFormat::PCR_Binary).value("CSV", PointCloudFormat::CSV)
        .value("LAS", PointCloudFormat::LAS).value("LAZ", PointCloudFormat::LAZ)
        .value("Auto", PointCloudFormat::Auto).export_values();

    py::class_<GeoTiffOptions>(m, "GeoTiffOptions")
        .def(py::init<>())
        .def_readwrite("cloud_optimized", &GeoTiffOptions::cloud_optimized)
        .def_readwrite("compress", &GeoTiffOptions::compress)
        .def_readwrite("compress_level", &GeoTiffOptions::compress_level)
        .def_readwrite("tile_width", &GeoTiffOptions::tile_width)
        .def_readwrite("tile_height", &GeoTiffOptions::tile_height)
        .def_readwrite("bigtiff", &GeoTiffOptions::bigtiff)
        .def_readwrite("overview_resampling", &GeoTiffOptions::overview_resampling);

    py::class_<PointCloudInfo>(m, "PointCloudInfo")
        .def(py::init<>())
        .def_readwrite("num_points", &PointCloudInfo::num_points)
        .def_readwrite("channels", &PointCloudInfo::channels)
        .def_readwrite("crs", &PointCloudInfo::crs)
        .def_readwrite("bounds", &PointCloudInfo::bounds);

    py::class_<PointCloudReader>(m, "PointCloudReader")
        .def_static("open", [](const std::string&, PointCloudFormat) -> PointCloudReader { unavailable("PointCloudReader.open"); },
                    py::arg("path"), py::arg("format") = PointCloudFormat::Auto);

    m.def("write_geotiff", [](const std::string&, const Grid&, const GridConfig&, const GeoTiffOptions&) {
        unavailable("write_geotiff");
    }, py::arg("path"), py::arg("grid"), py::arg("config"), py::arg("options") = GeoTiffOptions());
    m.def("read_geotiff_info", [](const std::string&) { unavailable("read_geotiff_info"); });
    m.def("read_point_cloud", [](const std::string&, PointCloudFormat) { unavailable("read_point_cloud"); },
          py::arg("path"), py::arg("format") = PointCloudFormat::Auto);
    m.def("write_point_cloud", [](const std::string&, const PointCloud&, PointCloudFormat) {
        unavailable("write_point_cloud");
    }, py::arg("path"), py::arg("cloud"), py::arg("format") = PointCloudFormat::PCR_Binary);
    m.def("read_point_cloud_info", [](const std::string&, PointCloudFormat) { unavailable("read_point_cloud_info"); },
          py::arg("path"), py::arg("format") = PointCloudFormat::Auto);
}
