// bind_engine.cpp -- FilterSpec, GlyphSpec, ReductionSpec, PipelineConfig, ProgressInfo, Pipeline.
#include "common.h"

#include "pcr/core/grid.h"
#include "pcr/core/point_cloud.h"
#include "pcr/engine/filter.h"
#include "pcr/engine/glyph.h"
#include "pcr/engine/pipeline.h"
#include "pcr/engine/sharded_pipeline.h"

using namespace pcr;

void bind_engine(py::module_& m) {
    py::enum_<ExecutionMode>(m, "ExecutionMode")
        .value("CPU", ExecutionMode::CPU).value("GPU", ExecutionMode::GPU)
        .value("Auto", ExecutionMode::Auto).value("Hybrid", ExecutionMode::Hybrid).export_values();

    py::enum_<CompareOp>(m, "CompareOp")
        .value("Equal", CompareOp::Equal).value("NotEqual", CompareOp::NotEqual)
        .value("Less", CompareOp::Less).value("LessEqual", CompareOp::LessEqual)
        .value("Greater", CompareOp::Greater).value("GreaterEqual", CompareOp::GreaterEqual)
        .value("InSet", CompareOp::InSet).value("NotInSet", CompareOp::NotInSet).export_values();

    py::enum_<GlyphType>(m, "GlyphType")
        .value("Point", GlyphType::Point).value("Line", GlyphType::Line)
        .value("Gaussian", GlyphType::Gaussian).export_values();

    py::class_<FilterPredicate>(m, "FilterPredicate")
        .def(py::init<>())
        .def_readwrite("channel_name", &FilterPredicate::channel_name)
        .def_readwrite("op", &FilterPredicate::op)
        .def_readwrite("value", &FilterPredicate::value)
        .def_readwrite("value_set", &FilterPredicate::value_set);

    py::class_<FilterSpec>(m, "FilterSpec")
        .def(py::init<>())
        .def_readwrite("predicates", &FilterSpec::predicates)
        .def("add", &FilterSpec::add, py::arg("channel"), py::arg("op"), py::arg("value"),
             py::return_value_policy::reference_internal)
        .def("add_in_set", &FilterSpec::add_in_set, py::arg("channel"), py::arg("values"),
             py::return_value_policy::reference_internal)
        .def("empty", &FilterSpec::empty);

    py::class_<GlyphSpec>(m, "GlyphSpec")
        .def(py::init<>())
        .def_readwrite("type", &GlyphSpec::type)
        .def_readwrite("direction_channel", &GlyphSpec::direction_channel)
        .def_readwrite("default_direction", &GlyphSpec::default_direction)
        .def_readwrite("half_length_channel", &GlyphSpec::half_length_channel)
        .def_readwrite("default_half_length", &GlyphSpec::default_half_length)
        .def_readwrite("sigma_x_channel", &GlyphSpec::sigma_x_channel)
        .def_readwrite("default_sigma_x", &GlyphSpec::default_sigma_x)
        .def_readwrite("sigma_y_channel", &GlyphSpec::sigma_y_channel)
        .def_readwrite("default_sigma_y", &GlyphSpec::default_sigma_y)
        .def_readwrite("rotation_channel", &GlyphSpec::rotation_channel)
        .def_readwrite("default_rotation", &GlyphSpec::default_rotation)
        .def_readwrite("max_radius_cells", &GlyphSpec::max_radius_cells)
        .def_readwrite("normalize_weights", &GlyphSpec::normalize_weights)
        .def("__repr__", [](const GlyphSpec& g) {
            static const char* names[] = {"Point", "Line", "Gaussian"};
            return std::string("GlyphSpec(type=") + names[static_cast<int>(g.type)] + ")";
        });

    py::class_<ReductionSpec>(m, "ReductionSpec")
        .def(py::init<>())
        .def_readwrite("value_channel", &ReductionSpec::value_channel)
        .def_readwrite("type", &ReductionSpec::type)
        .def_readwrite("weight_channel", &ReductionSpec::weight_channel)
        .def_readwrite("timestamp_channel", &ReductionSpec::timestamp_channel)
        .def_readwrite("percentile", &ReductionSpec::percentile)
        .def_readwrite("output_band_name", &ReductionSpec::output_band_name)
        .def_readwrite("glyph", &ReductionSpec::glyph);

    py::class_<PipelineConfig>(m, "PipelineConfig")
        .def(py::init<>())
        .def_readwrite("grid", &PipelineConfig::grid)
        .def_readwrite("reductions", &PipelineConfig::reductions)
        .def_readwrite("filter", &PipelineConfig::filter)
        .def_readwrite("target_crs", &PipelineConfig::target_crs)
        .def_readwrite("auto_reproject", &PipelineConfig::auto_reproject)
        .def_readwrite("exec_mode", &PipelineConfig::exec_mode)
        .def_readwrite("gpu_memory_budget", &PipelineConfig::gpu_memory_budget)
        .def_readwrite("host_cache_budget", &PipelineConfig::host_cache_budget)
        .def_readwrite("chunk_size", &PipelineConfig::chunk_size)
        .def_readwrite("cpu_threads", &PipelineConfig::cpu_threads)
        .def_readwrite("gpu_fallback_to_cpu", &PipelineConfig::gpu_fallback_to_cpu)
        .def_readwrite("hybrid_cpu_threads", &PipelineConfig::hybrid_cpu_threads)
        .def_readwrite("state_dir", &PipelineConfig::state_dir)
        .def_readwrite("resume", &PipelineConfig::resume)
        .def_readwrite("output_path", &PipelineConfig::output_path)
        .def_readwrite("write_cog", &PipelineConfig::write_cog)
        // fields the reference keeps C++-only, plus this build's extensions
        .def_readwrite("gpu_pool_size_bytes", &PipelineConfig::gpu_pool_size_bytes)
        .def_readwrite("cuda_device_id", &PipelineConfig::cuda_device_id)
        .def_readwrite("use_cuda_streams", &PipelineConfig::use_cuda_streams)
        .def_readwrite("gpu_require_strict", &PipelineConfig::gpu_require_strict)
        .def_readwrite("result_location", &PipelineConfig::result_location)
        .def_readwrite("shard_row_begin", &PipelineConfig::shard_row_begin)
        .def_readwrite("shard_row_end", &PipelineConfig::shard_row_end)
        .def_readwrite("shard_halo_rows", &PipelineConfig::shard_halo_rows)
        .def_readwrite("scatter_path", &PipelineConfig::scatter_path)
        .def_readwrite("finalize_with_first_ingest", &PipelineConfig::finalize_with_first_ingest);

    py::class_<ProgressInfo>(m, "ProgressInfo")
        .def(py::init<>())
        .def_readwrite("collections_processed", &ProgressInfo::collections_processed)
        .def_readwrite("collections_total", &ProgressInfo::collections_total)
        .def_readwrite("points_processed", &ProgressInfo::points_processed)
        .def_readwrite("tiles_active", &ProgressInfo::tiles_active)
        .def_readwrite("elapsed_seconds", &ProgressInfo::elapsed_seconds)
        .def("__repr__", [](const ProgressInfo& p) {
            return "ProgressInfo(points=" + std::to_string(p.points_processed) + ", tiles=" +
                   std::to_string(p.tiles_active) + ", elapsed=" + std::to_string(p.elapsed_seconds) + "s)";
        });

    py::class_<Pipeline>(m, "Pipeline")
        .def_static("create", &Pipeline::create)      // None on failure; reason: pcr.pipeline_create_error()
        .def("validate", [](const Pipeline& p) { raise_if_error(p.validate()); })
        .def("ingest", [](Pipeline& p, const PointCloud& c) { raise_if_error(p.ingest(c)); })
        .def("ingest_async", [](Pipeline& p, const PointCloud& c) { raise_if_error(p.ingest_async(c)); }, py::arg("cloud"),
             "ingest without waiting when the cloud is page-locked or device-resident (keep it alive until synchronize())")
        .def("ingest_file", [](Pipeline& p, const std::string& path, size_t chunk_points) {
            size_t n = 0;
            {
                py::gil_scoped_release release;
                raise_if_error(p.ingest_file(path, chunk_points, &n));
            }
            return n;
        }, py::arg("path"), py::arg("chunk_points") = size_t(4) << 20,
             "stream a PCRP / CSV file through page-locked double buffers; returns the number of points read")
        .def("finalize", [](Pipeline& p) { raise_if_error(p.finalize()); })
        .def("finalize_async", [](Pipeline& p) { raise_if_error(p.finalize_async()); },
             "Device-resident result: the finalize kernels are only enqueued on the pipeline's stream (complete after synchronize()); "
             "otherwise as finalize()")
        .def("run", [](Pipeline& p, const std::vector<const PointCloud*>& cs) { raise_if_error(p.run(cs)); })
        .def("set_progress_callback", &Pipeline::set_progress_callback)
        .def("result", &Pipeline::result, py::return_value_policy::reference_internal)
        .def("stats", &Pipeline::stats)
        .def("save_state", [](Pipeline& p, const std::string& dir) { raise_if_error(p.save_state(dir)); },
             py::arg("dir") = "")
        .def("load_state", [](Pipeline& p, const std::string& dir) { raise_if_error(p.load_state(dir)); },
             py::arg("dir") = "")
        // extensions for row-block sharded (multi-GPU) runs
        .def("halo_rows", &Pipeline::halo_rows)
        .def("line_reach_rows", [](Pipeline& p, const PointCloud& cloud) {
            int rows = 0;
            raise_if_error(p.line_reach_rows(cloud, &rows));
            return rows;
        })
        .def("state_row_begin", &Pipeline::state_row_begin)
        .def("state_row_count", &Pipeline::state_row_count)
        .def("state_planes", [](const Pipeline& p) {
            py::list out;
            for (const auto& v : p.state_planes())
                out.append(py::make_tuple(reinterpret_cast<uintptr_t>(v.device_ptr), v.plane_kind, v.group));
            return out;
        })
        .def("reduction_groups", &Pipeline::reduction_groups, "Accumulation group of every ReductionSpec (the group of state_planes())")
        .def("plane_reach_rows", [](const Pipeline& p) {
            py::list out;                       // aligned with state_planes(): 0 = the plane's halo rows stay empty (Point glyph)
            for (const auto& v : p.state_planes()) out.append(v.reach_rows);
            return out;
        })
        .def("tile_touched_ptr", [](const Pipeline& p, bool readonly) {
            int tx = 0, ty = 0;
            const void* d = readonly ? p.tile_touched_device_readonly(&tx, &ty) : p.tile_touched_device(&tx, &ty);
            return py::make_tuple(reinterpret_cast<uintptr_t>(d), tx, ty);
        }, py::arg("readonly") = false,
             "(device pointer, tiles_x, tiles_y) of the touched-tile flags; readonly=True promises not to write them (bands a "
             "scatter stored stay valid)")
        .def("merge_touched", [](Pipeline& p, uintptr_t d_union) {
            raise_if_error(p.merge_touched(reinterpret_cast<const void*>(d_union)));
        }, "OR the all-reduced touched flags of every rank (device words) into this pipeline's flags, on its stream")
        .def("result_band_device_ptr", [](const Pipeline& p, int band) { return reinterpret_cast<uintptr_t>(p.result_band_device(band)); },
             "Device address of finished band `band` (own rows x width floats) after finalize(); 0 when there is none")
        .def("synchronize", [](Pipeline& p) { raise_if_error(p.synchronize()); })
        .def("stream_ptr", [](const Pipeline& p) { return reinterpret_cast<uintptr_t>(p.stream_handle()); })
        .def("profile_enable", &Pipeline::profile_enable, py::arg("on"), py::arg("only_kernel") = "")
        .def("profile_read", [](Pipeline& p, bool reset) {
            py::dict d;
            for (const auto& k : p.profile_read(reset))
                d[py::str(k.name)] = py::make_tuple(k.launches, k.total_ms);
            return d;
        }, py::arg("reset") = true)
        .def("last_scatter", [](const Pipeline& p) {
            auto s = p.last_scatter();
            py::dict d;
            d["path"] = s.path == 2 ? "moments" : s.path == 1 ? "binned" : "direct";
            d["lds_tile"] = py::make_tuple(s.lds_tile_w, s.lds_tile_h);
            d["lds_apron"] = s.lds_apron;
            d["num_bins"] = s.num_bins;
            d["points_in"] = s.points_in;
            d["points_valid"] = s.points_valid;
            d["scatter_chunk"] = s.scatter_chunk;
            d["bands_with_scatter"] = s.bands_with_scatter;
            return d;
        })
        .def("engine", [](const Pipeline& p) { return std::string(p.engine()); },
             "'hip' (the MI355X engine) or 'host' (ExecutionMode.CPU, or a fallback the reference would have taken too)")
        .def("host_threads", &Pipeline::host_threads, "OpenMP threads of the host engine; 0 on the HIP engine")
        .def("spill_dir", &Pipeline::spill_dir, "Out of core: the directory of evicted bands (`.pcrt` tiles, reference layout); '' otherwise")
        .def("out_of_core", &Pipeline::out_of_core,
             "True when the grid's state exceeds gpu_memory_budget and the pipeline sweeps it in row bands of whole reference-tile rows");

    m.def("pipeline_create_error", &pipeline_create_error,
          "Why the last Pipeline.create() on this thread returned None");
    m.def("device_count", &cuda_device_count);
    m.def("device_name", &cuda_device_name, py::arg("device_id") = 0);

    // ---- the C++ ShardedPipeline (native RCCL exchange, pcr/engine/sharded_pipeline.h)
    py::class_<ShardedPipeline>(m, "NativeShardedPipeline")
        .def_static("make_id", []() {
            uint8_t id[ShardedPipeline::kIdBytes];
            raise_if_error(ShardedPipeline::make_id(id));
            return py::bytes(reinterpret_cast<const char*>(id), ShardedPipeline::kIdBytes);
        })
        .def_static("row_block", &ShardedPipeline::row_block, py::arg("rank"), py::arg("world"), py::arg("height"),
                    py::arg("align") = 1)
        .def_static("create", [](const PipelineConfig& cfg, const py::bytes& id, int rank, int world, int device, int align) {
            const std::string s = id;
            if (s.size() != (size_t)ShardedPipeline::kIdBytes) throw std::runtime_error("NativeShardedPipeline: the id must be 128 bytes");
            return ShardedPipeline::create(cfg, reinterpret_cast<const uint8_t*>(s.data()), rank, world, device, align);
        }, py::arg("cfg"), py::arg("id"), py::arg("rank"), py::arg("world"), py::arg("device"), py::arg("align") = 1)
        .def_static("create_error", &ShardedPipeline::create_error)
        .def("ingest", [](ShardedPipeline& p, const PointCloud& c) { raise_if_error(p.ingest(c)); })
        .def("exchange", [](ShardedPipeline& p) { raise_if_error(p.exchange()); })
        .def("finalize", [](ShardedPipeline& p) { raise_if_error(p.finalize()); })
        .def("save_state", [](ShardedPipeline& p, const std::string& dir) { raise_if_error(p.save_state(dir)); }, py::arg("dir") = "")
        .def("load_state", [](ShardedPipeline& p, const std::string& dir) { raise_if_error(p.load_state(dir)); }, py::arg("dir") = "")
        .def("ingest_unrouted", [](ShardedPipeline& p, const PointCloud& c) {
            size_t got = 0;
            raise_if_error(p.ingest_unrouted(c, &got));
            return got;
        }, "An arbitrary shard of the cloud: partitioned by owner on the device, exchanged (pcr_hip_comm_alltoallv), ingested; -> points received")
        .def("gather", [](ShardedPipeline& p, int dst_rank) {
            std::unique_ptr<Grid> g;
            raise_if_error(p.gather(dst_rank, &g));
            return g;
        }, py::arg("dst_rank") = 0, "Collective, after finalize(): the whole grid as one host Grid on dst_rank, None elsewhere")
        .def("result", &ShardedPipeline::result, py::return_value_policy::reference_internal)
        .def("pipeline", &ShardedPipeline::pipeline, py::return_value_policy::reference_internal)
        .def("row_begin", &ShardedPipeline::row_begin)
        .def("row_end", &ShardedPipeline::row_end)
        .def("halo_rows", &ShardedPipeline::halo_rows)
        .def("tiles_local", &ShardedPipeline::tiles_local)
        .def("bytes_sent", &ShardedPipeline::bytes_sent);
}
