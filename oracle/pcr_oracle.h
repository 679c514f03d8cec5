/*
 * pcr_oracle.h -- CPU restatement of the reference's Pipeline.ingest -> finalize path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported CPU baseline.  The product path
 * (pointcloud-raster_amd/) never links or calls it.
 *
 * Parity pin: validated against (a) the reference's own known-answer tests,
 * restated as data in tests/golden/reference_known_answers.json, and (b) the
 * reference's own accumulate / glyph / finalize code compiled unmodified into
 * oracle/_ref/libpcr_ref.so (see oracle/Makefile, oracle/ref_driver.cpp) and
 * run in the development container on seeded inputs (tests/test_oracle_vs_ref.py),
 * whose outputs are also committed as fixtures (tests/golden/ref_*.npz).
 *
 * All file:line citations are into the reference tree (BigHippo123/pointcloud-raster).
 *
 * Known, deliberate differences from the reference CPU engine:
 *   - no sort: points are folded into state in input order.  The reference sorts
 *     by (tile, cell) with an unstable std::sort first (src/engine/tile_router.cpp:138-240),
 *     so its per-cell fp32 summation order is implementation-defined; ours is input
 *     order.  Count/Min/Max are order independent (bit exact); Sum/Average differ
 *     only by fp32 re-association.
 *   - no LRU / disk spill (src/engine/tile_manager.cpp): every touched tile's state
 *     simply stays in memory.  "tile has state" == "a valid point's centre cell fell
 *     in it" (pipeline.cpp:688-691, 1220).
 *   - single threaded (reference results are only deterministic at cpu_threads=1).
 */
#ifndef PCR_ORACLE_H
#define PCR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same numbering as pcr::ReductionType (include/pcr/core/types.h:33-45). */
enum {
    PCRO_SUM = 0,
    PCRO_MAX = 1,
    PCRO_MIN = 2,
    PCRO_AVERAGE = 3,
    PCRO_WEIGHTED_AVERAGE = 4,
    PCRO_COUNT = 5
};

/* Same numbering as pcr::GlyphType (include/pcr/engine/glyph.h:11-15). */
enum { PCRO_GLYPH_POINT = 0, PCRO_GLYPH_LINE = 1, PCRO_GLYPH_GAUSSIAN = 2 };

/* Status codes, same numbering as pcr::StatusCode (types.h:118-126). */
enum {
    PCRO_OK = 0,
    PCRO_INVALID_ARGUMENT = 1,
    PCRO_OUT_OF_MEMORY = 2,
    PCRO_NOT_IMPLEMENTED = 6
};

/* The fields of pcr::GridConfig the hot path reads (grid_config.h:17-38). */
typedef struct pcro_grid {
    double min_x, min_y, max_x, max_y;
    double cell_size_x, cell_size_y;
    int32_t width, height;
    int32_t tile_width, tile_height;
} pcro_grid;

/* The fields of pcr::GlyphSpec the kernels read (glyph.h:20-42). */
typedef struct pcro_glyph {
    int32_t type;
    float default_direction;
    float default_half_length;
    float default_sigma_x;
    float default_sigma_y;
    float default_rotation;
    float max_radius_cells;
} pcro_glyph;

/* One batch of points, SoA.  Optional per-point glyph arrays may be NULL
 * (-> GlyphSpec default), like TileBatch (tile_router.h:37-55). */
typedef struct pcro_points {
    const double* x;
    const double* y;
    const float* value;
    const float* direction;
    const float* half_length;
    const float* sigma_x;
    const float* sigma_y;
    const float* rotation;
    uint64_t n;
} pcro_points;

/* ---- value-type helpers (src/core/grid_config.cpp) ------------------------ */
void pcro_compute_dimensions(pcro_grid* g, int32_t* tiles_x, int32_t* tiles_y);
int pcro_world_to_cell(const pcro_grid* g, double wx, double wy, int32_t* col, int32_t* row);
void pcro_tile_cell_range(const pcro_grid* g, int32_t tile_row, int32_t tile_col,
                          int32_t* col_start, int32_t* row_start,
                          int32_t* col_count, int32_t* row_count);

/* ---- op algebra on a band-sequential tile state (builtin_ops.h, reduction_registry.cpp) */
int pcro_state_floats(int rtype);                                 /* 0 if unregistered */
int pcro_init_state(int rtype, float* state, int64_t tile_cells);
int pcro_accumulate(int rtype, const uint32_t* cell_indices, const float* values,
                    float* state, size_t num_points, int64_t tile_cells);
int pcro_merge_state(int rtype, float* dst, const float* src, int64_t tile_cells);
int pcro_finalize_state(int rtype, const float* state, float* out, int64_t tile_cells);

/* ---- glyph splat into ONE tile's state (src/engine/glyph_kernels.cu:571-604) */
int pcro_accumulate_glyph(const pcro_glyph* spec, int rtype, const pcro_points* batch,
                          float* state, int64_t tile_cells, const pcro_grid* g,
                          int32_t tile_col_origin, int32_t tile_row_origin,
                          int32_t tile_w, int32_t tile_h);

/* ---- point filter (src/engine/filter.cpp:34-56 evaluate_predicate, :141-206 AND over predicates) ---- */
enum { PCRO_CMP_EQUAL = 0, PCRO_CMP_NOT_EQUAL, PCRO_CMP_LESS, PCRO_CMP_LESS_EQUAL, PCRO_CMP_GREATER,
       PCRO_CMP_GREATER_EQUAL, PCRO_CMP_IN_SET, PCRO_CMP_NOT_IN_SET };   /* pcr::CompareOp, filter.h:20-29 */
typedef struct pcro_predicate {
    const float* channel;
    int32_t op;
    float value;
    const float* set;
    int32_t set_size;
} pcro_predicate;
/* mask[i] = 1 where every predicate holds; returns the number of survivors */
uint64_t pcro_filter_mask(const pcro_predicate* preds, int n_pred, uint64_t n, uint8_t* mask);

/* ---- the whole path for one ReductionSpec ---------------------------------- */
typedef struct pcro_reduction pcro_reduction;

/* wide != 0: keep the state in double and add exact products; NOT reference
 * behaviour, only used by tests to bound fp32 re-association error. */
pcro_reduction* pcro_create(const pcro_grid* g, int rtype, const pcro_glyph* glyph, int wide);
void pcro_destroy(pcro_reduction* r);
int pcro_ingest(pcro_reduction* r, const pcro_points* pts);       /* process_cloud */
int pcro_finalize(const pcro_reduction* r, float* band);          /* finalize_result, W*H floats */
/* test hooks */
int pcro_tile_touched(const pcro_reduction* r, int32_t tile_row, int32_t tile_col);
uint64_t pcro_points_valid(const pcro_reduction* r);
const char* pcro_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
