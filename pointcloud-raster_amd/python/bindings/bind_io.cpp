// bind_io.cpp -- the reference's I/O layer at the same Python names (python/bindings.cpp:500-640):
// PCRP / CSV point clouds, GeoTIFF output of finalized grids, `.pcrt` tile-state checkpoints.
// LAS/LAZ raise NotImplemented as upstream.
#include "common.h"

#include "pcr/core/grid.h"
#include "pcr/core/grid_config.h"
#include "pcr/core/point_cloud.h"
#include "pcr/io/grid_io.h"
#include "pcr/io/point_cloud_io.h"
#include "pcr/io/tile_state_io.h"

#include <vector>

using namespace pcr;

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
        .def_static("open", [](const std::string& path, PointCloudFormat format) {
            auto r = PointCloudReader::open(path, format);
            if (!r) throw std::runtime_error("Failed to open point cloud: " + path);
            return r;
        }, py::arg("path"), py::arg("format") = PointCloudFormat::Auto)
        .def("info", &PointCloudReader::info, py::return_value_policy::reference_internal)
        .def("read_chunk", &PointCloudReader::read_chunk, py::arg("cloud"), py::arg("max_points"))
        .def("rewind", [](PointCloudReader& r) { raise_if_error(r.rewind()); })
        .def("eof", &PointCloudReader::eof);

    // extension (the reference does not bind it): incremental assembly from reference tiles
    py::class_<TiledGeoTiffWriter>(m, "TiledGeoTiffWriter")
        .def_static("open", [](const std::string& path, const GridConfig& config, const std::vector<std::string>& band_names,
                               const GeoTiffOptions& options) {
            auto w = TiledGeoTiffWriter::open(path, config, band_names, options);
            if (!w) throw std::runtime_error("TiledGeoTiffWriter.open: failed to create " + path);
            return w;
        }, py::arg("path"), py::arg("config"), py::arg("band_names"), py::arg("options") = GeoTiffOptions())
        .def("write_tile", [](TiledGeoTiffWriter& w, int tile_row, int tile_col,
                              py::array_t<float, py::array::c_style | py::array::forcecast> data) {
            auto b = data.request();
            if (b.ndim != 3) throw std::runtime_error("write_tile: data must be [bands, rows, cols]");
            TileIndex t;
            t.row = tile_row;
            t.col = tile_col;
            raise_if_error(w.write_tile(t, static_cast<const float*>(b.ptr), (int)b.shape[0]));
        }, py::arg("tile_row"), py::arg("tile_col"), py::arg("data"))
        .def("close", [](TiledGeoTiffWriter& w) { raise_if_error(w.close()); });

    m.def("write_geotiff", [](const std::string& path, const Grid& grid, const GridConfig& config, const GeoTiffOptions& options) {
        raise_if_error(write_geotiff(path, grid, config, options));
    }, py::arg("path"), py::arg("grid"), py::arg("config"), py::arg("options") = GeoTiffOptions());
    m.def("read_geotiff_info", [](const std::string& path) {
        int w = 0, h = 0, nb = 0;
        CRS crs;
        BBox bounds;
        raise_if_error(read_geotiff_info(path, w, h, nb, crs, bounds));
        return py::make_tuple(w, h, nb, crs, bounds);
    }, py::arg("path"));
    m.def("read_geotiff_band", [](const std::string& path, int band_index) {
        int w = 0, h = 0, nb = 0;
        CRS crs;
        BBox bounds;
        raise_if_error(read_geotiff_info(path, w, h, nb, crs, bounds));
        py::array_t<float> out({h, w});
        raise_if_error(read_geotiff_band(path, band_index, out.mutable_data(), w, h));
        return out;
    }, py::arg("path"), py::arg("band_index") = 0);
    m.def("read_geotiff_band_names", [](const std::string& path) {
        std::vector<std::string> names;
        raise_if_error(read_geotiff_band_names(path, names));
        return names;
    }, py::arg("path"));

    m.def("read_point_cloud", [](const std::string& path, PointCloudFormat format, MemoryLocation location) {
        auto c = read_point_cloud(path, format, location);
        if (!c) throw std::runtime_error("Failed to read point cloud: " + path);
        return c;
    }, py::arg("path"), py::arg("format") = PointCloudFormat::Auto, py::arg("location") = MemoryLocation::Host);
    m.def("write_point_cloud", [](const std::string& path, const PointCloud& cloud, PointCloudFormat format) {
        raise_if_error(write_point_cloud(path, cloud, format));
    }, py::arg("path"), py::arg("cloud"), py::arg("format") = PointCloudFormat::PCR_Binary);
    m.def("read_point_cloud_info", [](const std::string& path, PointCloudFormat format) {
        PointCloudInfo info;
        raise_if_error(read_point_cloud_info(path, info, format));
        return info;
    }, py::arg("path"), py::arg("format") = PointCloudFormat::Auto);
}
