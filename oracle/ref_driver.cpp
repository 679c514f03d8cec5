// ref_driver.cpp -- C-ABI harness over the REFERENCE's own CPU code, test infrastructure only.
//
// Compiled (oracle/Makefile, target `ref`) together with the reference's source files
// where they lie under $PCR_REFERENCE_DIR (default /root/reference):
//     src/engine/glyph_kernels.cu   (as C++, exactly as the reference's CPU build does,
//                                    CMakeLists.txt:162-165)
//     src/ops/reduction_registry.cpp
//     src/engine/accumulator.cpp
//     src/io/tile_state_io.cpp      (the `.pcrt` checkpoint writer / reader)
// into oracle/_ref/libpcr_ref.so.  Those files link without anything the image
// lacks.  The rest of the reference pipeline (src/core/types.cpp -> <proj.h>,
// src/io/grid_io.cpp -> GDAL, src/engine/pipeline.cpp -> both + memory_pool.cu) would
// need stand-in headers/libraries and is therefore treated as unbuildable here; its
// routing/tiling/finalize-assembly behaviour is pinned by the reference's own
// known-answer tests instead (tests/golden/reference_known_answers.json).
//
// This file contains no reference code: it only calls pcr::get_reduction(),
// pcr::Accumulator and pcr::accumulate_glyph() through their public headers.
// Only exists in the development container; never shipped, never on the GPU box
// except as the prebuilt oracle/_ref/*.so used by tests as a checker.

#include "pcr/core/grid_config.h"
#include "pcr/engine/accumulator.h"
#include "pcr/engine/glyph.h"
#include "pcr/engine/glyph_kernels.h"
#include "pcr/engine/tile_router.h"
#include "pcr/io/tile_state_io.h"
#include "pcr/ops/reduction_registry.h"

#include "pcr_oracle.h"

#include <cstdio>
#include <cstring>
#include <string>

namespace {
thread_local std::string g_msg;

int ret(const pcr::Status& s) {
    g_msg = s.message;
    return static_cast<int>(s.code);
}

pcr::GridConfig to_cfg(const pcro_grid* g) {
    pcr::GridConfig c;
    c.bounds.min_x = g->min_x;
    c.bounds.min_y = g->min_y;
    c.bounds.max_x = g->max_x;
    c.bounds.max_y = g->max_y;
    c.cell_size_x = g->cell_size_x;
    c.cell_size_y = g->cell_size_y;
    c.width = g->width;
    c.height = g->height;
    c.tile_width = g->tile_width;
    c.tile_height = g->tile_height;
    return c;
}
}  // namespace

extern "C" {

const char* pcr_ref_last_error(void) { return g_msg.c_str(); }

int pcr_ref_state_floats(int rtype) {
    const pcr::ReductionInfo* info = pcr::get_reduction(static_cast<pcr::ReductionType>(rtype));
    return info ? info->state_floats : 0;
}

int pcr_ref_init_state(int rtype, float* state, int64_t cells) {
    const pcr::ReductionInfo* info = pcr::get_reduction(static_cast<pcr::ReductionType>(rtype));
    if (!info) return 1;
    return ret(info->init_state(state, cells, nullptr));
}

// Through pcr::Accumulator (src/engine/accumulator.cpp), the call process_cloud makes.
int pcr_ref_accumulate(int rtype, const uint32_t* cells, const float* values, float* state,
                       size_t n, int64_t tile_cells) {
    auto acc = pcr::Accumulator::create(nullptr);
    if (!acc) return 2;
    pcr::TileBatch b;
    b.local_cell_indices = const_cast<uint32_t*>(cells);
    b.values = const_cast<float*>(values);
    b.num_points = n;
    return ret(acc->accumulate(static_cast<pcr::ReductionType>(rtype), b, state, tile_cells));
}

int pcr_ref_merge_state(int rtype, float* dst, const float* src, int64_t cells) {
    const pcr::ReductionInfo* info = pcr::get_reduction(static_cast<pcr::ReductionType>(rtype));
    if (!info) return 1;
    return ret(info->merge_state(dst, src, cells, nullptr));
}

int pcr_ref_finalize_state(int rtype, const float* state, float* out, int64_t cells) {
    const pcr::ReductionInfo* info = pcr::get_reduction(static_cast<pcr::ReductionType>(rtype));
    if (!info) return 1;
    return ret(info->finalize(state, out, cells, nullptr));
}

int pcr_ref_accumulate_glyph(const pcro_glyph* spec, int rtype, const pcro_points* pts,
                             float* state, int64_t tile_cells, const pcro_grid* g,
                             int32_t tile_col_origin, int32_t tile_row_origin,
                             int32_t tile_w, int32_t tile_h) {
    pcr::GlyphSpec gs;
    gs.type = static_cast<pcr::GlyphType>(spec->type);
    gs.default_direction = spec->default_direction;
    gs.default_half_length = spec->default_half_length;
    gs.default_sigma_x = spec->default_sigma_x;
    gs.default_sigma_y = spec->default_sigma_y;
    gs.default_rotation = spec->default_rotation;
    gs.max_radius_cells = spec->max_radius_cells;

    pcr::TileBatch b;
    b.values = const_cast<float*>(pts->value);
    b.num_points = pts->n;
    b.coord_x = const_cast<double*>(pts->x);
    b.coord_y = const_cast<double*>(pts->y);
    b.glyph_direction = const_cast<float*>(pts->direction);
    b.glyph_half_length = const_cast<float*>(pts->half_length);
    b.glyph_sigma_x = const_cast<float*>(pts->sigma_x);
    b.glyph_sigma_y = const_cast<float*>(pts->sigma_y);
    b.glyph_rotation = const_cast<float*>(pts->rotation);

    pcr::GridConfig cfg = to_cfg(g);
    return ret(pcr::accumulate_glyph(gs, static_cast<pcr::ReductionType>(rtype), b, state,
                                     tile_cells, cfg, tile_col_origin, tile_row_origin,
                                     tile_w, tile_h, nullptr));
}

// src/io/tile_state_io.cpp: the reference's own `.pcrt` writer / reader.
int pcr_ref_write_tile_state(const char* path, int tile_row, int tile_col, int cols, int rows, int state_floats,
                             int rtype, const float* state) {
    pcr::TileIndex t;
    t.row = tile_row;
    t.col = tile_col;
    return ret(pcr::write_tile_state(path, t, cols, rows, state_floats, static_cast<pcr::ReductionType>(rtype), state));
}

int pcr_ref_read_tile_state(const char* path, int* tile_row, int* tile_col, int* cols, int* rows, int* state_floats,
                            int* rtype, float* state /* may be null: header only */) {
    pcr::TileIndex t;
    pcr::ReductionType type;
    pcr::Status s = state ? pcr::read_tile_state(path, t, *cols, *rows, *state_floats, type, state)
                          : pcr::read_tile_state_header(path, t, *cols, *rows, *state_floats, type);
    if (s.ok()) {
        *tile_row = t.row;
        *tile_col = t.col;
        *rtype = static_cast<int>(type);
    }
    return ret(s);
}

void pcr_ref_tile_state_filename(const char* dir, int tile_row, int tile_col, char* out, int cap) {
    pcr::TileIndex t;
    t.row = tile_row;
    t.col = tile_col;
    std::string p = pcr::tile_state_filename(dir, t);
    std::snprintf(out, cap, "%s", p.c_str());
}

}  // extern "C"
