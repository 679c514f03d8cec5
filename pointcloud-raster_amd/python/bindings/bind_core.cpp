// bind_core.cpp -- enums, BBox/CRS/..., GridConfig, Grid, PointCloud.
#include "common.h"

#include "pcr/core/grid.h"
#include "pcr/core/grid_config.h"
#include "pcr/core/point_cloud.h"

#include <cstring>

using namespace pcr;

namespace {

std::string num(double v) { return std::to_string(v); }

template <typename T>
py::array_t<T> host_view(T* data, size_t n, py::handle owner, const char* what) {
    if (!data) throw std::runtime_error(std::string(what) + ": no data");
    return py::array_t<T>({static_cast<py::ssize_t>(n)}, {static_cast<py::ssize_t>(sizeof(T))}, data, owner);
}

void require_host(MemoryLocation loc, const char* what) {
    if (loc == MemoryLocation::Device)
        throw std::runtime_error(std::string(what) + ": data lives in Device memory; call to_host() first");
}

template <typename T>
void copy_in(T* dst, const py::array_t<T, py::array::c_style | py::array::forcecast>& arr, size_t limit,
             const char* too_big) {
    auto buf = arr.request();
    if (buf.ndim != 1) throw std::runtime_error("expected a 1-D array");
    if (static_cast<size_t>(buf.shape[0]) > limit) throw std::runtime_error(too_big);
    std::memcpy(dst, buf.ptr, static_cast<size_t>(buf.shape[0]) * sizeof(T));
}

}  // namespace

void bind_core(py::module_& m) {
    py::enum_<DataType>(m, "DataType")
        .value("Float32", DataType::Float32).value("Float64", DataType::Float64)
        .value("Int32", DataType::Int32).value("UInt32", DataType::UInt32)
        .value("Int16", DataType::Int16).value("UInt16", DataType::UInt16)
        .value("UInt8", DataType::UInt8).export_values();

    py::enum_<ReductionType>(m, "ReductionType")
        .value("Sum", ReductionType::Sum).value("Max", ReductionType::Max).value("Min", ReductionType::Min)
        .value("Average", ReductionType::Average).value("WeightedAverage", ReductionType::WeightedAverage)
        .value("Count", ReductionType::Count).value("Median", ReductionType::Median)
        .value("Percentile", ReductionType::Percentile).value("MostRecent", ReductionType::MostRecent)
        .value("PriorityMerge", ReductionType::PriorityMerge).value("Custom", ReductionType::Custom)
        .export_values();

    py::enum_<MemoryLocation>(m, "MemoryLocation")
        .value("Host", MemoryLocation::Host).value("HostPinned", MemoryLocation::HostPinned)
        .value("Device", MemoryLocation::Device).export_values();

    py::enum_<StatusCode>(m, "StatusCode")
        .value("Ok", StatusCode::Ok).value("InvalidArgument", StatusCode::InvalidArgument)
        .value("OutOfMemory", StatusCode::OutOfMemory).value("CudaError", StatusCode::CudaError)
        .value("IoError", StatusCode::IoError).value("CrsError", StatusCode::CrsError)
        .value("NotImplemented", StatusCode::NotImplemented).export_values();

    py::class_<BBox>(m, "BBox")
        .def(py::init<>())
        .def(py::init([](double x0, double y0, double x1, double y1) {
                 BBox b; b.min_x = x0; b.min_y = y0; b.max_x = x1; b.max_y = y1; return b; }),
             py::arg("min_x"), py::arg("min_y"), py::arg("max_x"), py::arg("max_y"))
        .def_readwrite("min_x", &BBox::min_x).def_readwrite("min_y", &BBox::min_y)
        .def_readwrite("max_x", &BBox::max_x).def_readwrite("max_y", &BBox::max_y)
        .def("expand", py::overload_cast<double, double>(&BBox::expand))
        .def("expand", py::overload_cast<const BBox&>(&BBox::expand))
        .def("contains", &BBox::contains)
        .def("width", &BBox::width).def("height", &BBox::height).def("valid", &BBox::valid)
        .def("__repr__", [](const BBox& b) {
            return "BBox(min_x=" + num(b.min_x) + ", min_y=" + num(b.min_y) +
                   ", max_x=" + num(b.max_x) + ", max_y=" + num(b.max_y) + ")";
        });

    py::class_<CRS>(m, "CRS")
        .def(py::init<>())
        .def_readwrite("wkt", &CRS::wkt).def_readwrite("epsg", &CRS::epsg)
        .def("is_projected", &CRS::is_projected).def("is_geographic", &CRS::is_geographic)
        .def("is_valid", &CRS::is_valid)
        .def_static("from_epsg", &CRS::from_epsg).def_static("from_wkt", &CRS::from_wkt)
        .def("equivalent_to", &CRS::equivalent_to)
        .def("__repr__", [](const CRS& c) {
            return c.epsg ? "CRS(epsg=" + std::to_string(c.epsg) + ")" : "CRS(wkt='" + c.wkt.substr(0, 50) + "...')";
        });

    py::class_<NoDataPolicy>(m, "NoDataPolicy")
        .def(py::init<>())
        .def_readwrite("value", &NoDataPolicy::value).def_readwrite("use_nan", &NoDataPolicy::use_nan)
        .def("sentinel", &NoDataPolicy::sentinel);

    py::class_<TileIndex>(m, "TileIndex")
        .def(py::init<>())
        .def(py::init([](int row, int col) { TileIndex t; t.row = row; t.col = col; return t; }))
        .def_readwrite("row", &TileIndex::row).def_readwrite("col", &TileIndex::col)
        .def("__eq__", &TileIndex::operator==).def("__lt__", &TileIndex::operator<)
        .def("__repr__", [](const TileIndex& t) {
            return "TileIndex(row=" + std::to_string(t.row) + ", col=" + std::to_string(t.col) + ")";
        });

    py::class_<Status>(m, "Status")
        .def(py::init<>())
        .def_readwrite("code", &Status::code).def_readwrite("message", &Status::message)
        .def("ok", &Status::ok)
        .def_static("success", &Status::success).def_static("error", &Status::error)
        .def("__bool__", &Status::ok)
        .def("__repr__", [](const Status& s) -> std::string {
            if (s.ok()) return "Status(Ok)";
            return "Status(code=" + std::to_string(static_cast<int>(s.code)) + ", message='" + s.message + "')";
        });

    py::class_<ChannelDesc>(m, "ChannelDesc")
        .def(py::init<>())
        .def_readwrite("name", &ChannelDesc::name).def_readwrite("dtype", &ChannelDesc::dtype)
        .def_readwrite("offset", &ChannelDesc::offset);

    py::class_<BandDesc>(m, "BandDesc")
        .def(py::init<>())
        .def_readwrite("name", &BandDesc::name).def_readwrite("dtype", &BandDesc::dtype)
        .def_readwrite("is_state", &BandDesc::is_state);

    py::class_<GridConfig>(m, "GridConfig")
        .def(py::init<>())
        .def_readwrite("bounds", &GridConfig::bounds).def_readwrite("crs", &GridConfig::crs)
        .def_readwrite("cell_size_x", &GridConfig::cell_size_x).def_readwrite("cell_size_y", &GridConfig::cell_size_y)
        .def_readwrite("width", &GridConfig::width).def_readwrite("height", &GridConfig::height)
        .def_readwrite("nodata", &GridConfig::nodata)
        .def_readwrite("tile_width", &GridConfig::tile_width).def_readwrite("tile_height", &GridConfig::tile_height)
        .def_readwrite("tiles_x", &GridConfig::tiles_x).def_readwrite("tiles_y", &GridConfig::tiles_y)
        .def("compute_dimensions", &GridConfig::compute_dimensions)
        .def("world_to_cell", [](const GridConfig& g, double wx, double wy) {
            int col = 0, row = 0;
            bool ok = g.world_to_cell(wx, wy, col, row);
            return py::make_tuple(col, row, ok);
        })
        .def("cell_to_world", [](const GridConfig& g, int col, int row) {
            double wx = 0, wy = 0;
            g.cell_to_world(col, row, wx, wy);
            return py::make_tuple(wx, wy);
        })
        .def("cell_to_tile", &GridConfig::cell_to_tile)
        .def("tile_bounds", &GridConfig::tile_bounds)
        .def("tile_cell_range", [](const GridConfig& g, TileIndex t) {
            int c0 = 0, r0 = 0, nc = 0, nr = 0;
            g.tile_cell_range(t, c0, r0, nc, nr);
            return py::make_tuple(c0, r0, nc, nr);
        })
        .def("total_tiles", &GridConfig::total_tiles).def("total_cells", &GridConfig::total_cells)
        .def("gdal_geotransform", [](const GridConfig& g) {       // C++-only in the reference; bound for the known-answer tests
            double gt[6];
            g.gdal_geotransform(gt);
            return py::make_tuple(gt[0], gt[1], gt[2], gt[3], gt[4], gt[5]);
        })
        .def("validate", [](const GridConfig& g) { raise_if_error(g.validate()); })
        .def("__repr__", [](const GridConfig& g) {
            return "GridConfig(width=" + std::to_string(g.width) + ", height=" + std::to_string(g.height) +
                   ", tiles=" + std::to_string(g.tiles_x) + "x" + std::to_string(g.tiles_y) + ")";
        });

    py::class_<Grid>(m, "Grid")
        .def_static("create", &Grid::create, py::arg("cols"), py::arg("rows"), py::arg("bands"),
                    py::arg("loc") = MemoryLocation::Host)
        .def_static("create_for_tile", &Grid::create_for_tile, py::arg("config"), py::arg("tile"),
                    py::arg("bands"), py::arg("loc") = MemoryLocation::Host)
        .def("num_bands", &Grid::num_bands).def("band_desc", &Grid::band_desc)
        .def("band_index", &Grid::band_index)
        .def("cols", &Grid::cols).def("rows", &Grid::rows).def("cell_count", &Grid::cell_count)
        .def("location", &Grid::location)
        .def("fill", [](Grid& g, float v) { raise_if_error(g.fill(v)); })
        .def("fill_band", [](Grid& g, int i, float v) { raise_if_error(g.fill_band(i, v)); })
        .def("band_array", [](Grid& g, int i) {
            float* p = g.band_f32(i);
            if (!p) throw std::runtime_error("Invalid band index or data type");
            require_host(g.location(), "band_array");
            return py::array_t<float>({g.rows(), g.cols()},
                                      {static_cast<py::ssize_t>(g.cols() * sizeof(float)),
                                       static_cast<py::ssize_t>(sizeof(float))},
                                      p, py::cast(&g));
        })
        .def("set_band_array", [](Grid& g, int i, py::array_t<float, py::array::c_style | py::array::forcecast> a) {
            float* p = g.band_f32(i);
            if (!p) throw std::runtime_error("Invalid band index or data type");
            require_host(g.location(), "set_band_array");
            auto buf = a.request();
            if (buf.ndim != 2 || buf.shape[0] != g.rows() || buf.shape[1] != g.cols())
                throw std::runtime_error("Array shape mismatch");
            std::memcpy(p, buf.ptr, static_cast<size_t>(g.cell_count()) * sizeof(float));
        })
        // extensions: device-resident results
        .def("band_device_ptr", [](Grid& g, int i) {
            if (g.location() != MemoryLocation::Device) throw std::runtime_error("band_device_ptr: grid is not on Device");
            float* p = g.band_f32(i);
            if (!p) throw std::runtime_error("Invalid band index or data type");
            return reinterpret_cast<uintptr_t>(p);
        })
        .def("to_host", [](const Grid& g) {
            auto h = g.to(MemoryLocation::Host);
            if (!h) throw std::runtime_error("Failed to copy grid to Host memory");
            return h;
        })
        .def("__repr__", [](const Grid& g) {
            return "Grid(cols=" + std::to_string(g.cols()) + ", rows=" + std::to_string(g.rows()) +
                   ", bands=" + std::to_string(g.num_bands()) + ")";
        });

    py::class_<PointCloud>(m, "PointCloud")
        .def_static("create", &PointCloud::create, py::arg("capacity"), py::arg("loc") = MemoryLocation::Host)
        .def("add_channel", [](PointCloud& pc, const std::string& name, DataType dt) {
            raise_if_error(pc.add_channel(name, dt));
        }, py::arg("name"), py::arg("dtype") = DataType::Float32)
        .def("has_channel", &PointCloud::has_channel)
        .def("channel", &PointCloud::channel, py::return_value_policy::reference_internal)
        .def("channel_names", &PointCloud::channel_names)
        .def("count", &PointCloud::count).def("capacity", &PointCloud::capacity)
        .def("location", &PointCloud::location)
        .def("crs", &PointCloud::crs).def("set_crs", &PointCloud::set_crs)
        .def("resize", [](PointCloud& pc, size_t n) { raise_if_error(pc.resize(n)); })
        .def("x_array", [](PointCloud& pc) {
            require_host(pc.location(), "x_array");
            return host_view<double>(pc.x(), pc.count(), py::cast(&pc), "x_array");
        })
        .def("y_array", [](PointCloud& pc) {
            require_host(pc.location(), "y_array");
            return host_view<double>(pc.y(), pc.count(), py::cast(&pc), "y_array");
        })
        .def("channel_array_f32", [](PointCloud& pc, const std::string& name) {
            float* p = pc.channel_f32(name);
            if (!p) throw std::runtime_error("Channel not found or wrong type: " + name);
            require_host(pc.location(), "channel_array_f32");
            return host_view<float>(p, pc.count(), py::cast(&pc), "channel_array_f32");
        })
        // set_x_array also sets the point count (len(arr)); set_y_array only copies -- reference behaviour
        .def("set_x_array", [](PointCloud& pc, py::array_t<double, py::array::c_style | py::array::forcecast> a) {
            require_host(pc.location(), "set_x_array");
            copy_in<double>(pc.x(), a, pc.capacity(), "Array too large for capacity");
            raise_if_error(pc.resize(static_cast<size_t>(a.request().shape[0])));
        })
        .def("set_y_array", [](PointCloud& pc, py::array_t<double, py::array::c_style | py::array::forcecast> a) {
            require_host(pc.location(), "set_y_array");
            copy_in<double>(pc.y(), a, pc.capacity(), "Array too large for capacity");
        })
        .def("set_channel_array_f32", [](PointCloud& pc, const std::string& name,
                                         py::array_t<float, py::array::c_style | py::array::forcecast> a) {
            float* p = pc.channel_f32(name);
            if (!p) throw std::runtime_error("Channel not found or wrong type: " + name);
            require_host(pc.location(), "set_channel_array_f32");
            copy_in<float>(p, a, pc.count(), "Array size exceeds point count");
        })
        // extension: any channel as a zero-copy array of its own dtype (files carry f64 / i32 / u32 channels too)
        .def("channel_array", [](PointCloud& pc, const std::string& name) -> py::object {
            const ChannelDesc* d = pc.channel(name);
            void* p = pc.channel_data(name);
            if (!d || !p) throw std::runtime_error("Channel not found: " + name);
            require_host(pc.location(), "channel_array");
            py::object owner = py::cast(&pc);
            switch (d->dtype) {
                case DataType::Float32: return host_view<float>(static_cast<float*>(p), pc.count(), owner, "channel_array");
                case DataType::Float64: return host_view<double>(static_cast<double*>(p), pc.count(), owner, "channel_array");
                case DataType::Int32: return host_view<int32_t>(static_cast<int32_t*>(p), pc.count(), owner, "channel_array");
                case DataType::UInt32: return host_view<uint32_t>(static_cast<uint32_t*>(p), pc.count(), owner, "channel_array");
                case DataType::Int16: return host_view<int16_t>(static_cast<int16_t*>(p), pc.count(), owner, "channel_array");
                case DataType::UInt16: return host_view<uint16_t>(static_cast<uint16_t*>(p), pc.count(), owner, "channel_array");
                default: return host_view<uint8_t>(static_cast<uint8_t*>(p), pc.count(), owner, "channel_array");
            }
        })
        .def("to_device", [](const PointCloud& pc) {
            auto d = pc.to(MemoryLocation::Device);
            if (!d) throw std::runtime_error("Failed to transfer point cloud to Device memory "
                                             "(HIP out of memory or no usable GPU)");
            return d;
        }, "Copy the cloud (coordinates and every channel) into GPU memory")
        .def("to_host", [](const PointCloud& pc) {
            auto h = pc.to(MemoryLocation::Host);
            if (!h) throw std::runtime_error("Failed to transfer point cloud to Host memory");
            return h;
        }, "Copy the cloud into Host memory")
        // extension: raw device addresses for zero-copy interop (torch / __cuda_array_interface__)
        .def("device_ptrs", [](PointCloud& pc) {
            py::dict d;
            d["x"] = reinterpret_cast<uintptr_t>(pc.x());
            d["y"] = reinterpret_cast<uintptr_t>(pc.y());
            for (const auto& n : pc.channel_names()) d[py::str(n)] = reinterpret_cast<uintptr_t>(pc.channel_data(n));
            return d;
        })
        .def("__repr__", [](const PointCloud& pc) {
            return "PointCloud(count=" + std::to_string(pc.count()) + ", capacity=" + std::to_string(pc.capacity()) +
                   ", channels=" + std::to_string(pc.channel_names().size()) + ")";
        });
}
