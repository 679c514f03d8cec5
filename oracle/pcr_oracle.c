/*
 * pcr_oracle.c -- CPU restatement of the reference's ingest -> finalize path.
 * TEST INFRASTRUCTURE ONLY; see pcr_oracle.h for the rules and the parity pin.
 * Citations (file:line) are into the reference tree.
 */
#include "pcr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];

static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

const char* pcro_last_error(void) { return g_err; }

/* ------------------------------------------------------------------------- */
/* GridConfig (src/core/grid_config.cpp)                                      */
/* ------------------------------------------------------------------------- */

/* grid_config.cpp:7-22 */
void pcro_compute_dimensions(pcro_grid* g, int32_t* tiles_x, int32_t* tiles_y) {
    int valid = g->max_x >= g->min_x && g->max_y >= g->min_y;   /* types.h:63 */
    if (!valid) {
        g->width = g->height = 0;
        if (tiles_x) *tiles_x = 0;
        if (tiles_y) *tiles_y = 0;
        return;
    }
    g->width = (int32_t)ceil((g->max_x - g->min_x) / fabs(g->cell_size_x));
    g->height = (int32_t)ceil((g->max_y - g->min_y) / fabs(g->cell_size_y));
    if (tiles_x) *tiles_x = (g->width + g->tile_width - 1) / g->tile_width;
    if (tiles_y) *tiles_y = (g->height + g->tile_height - 1) / g->tile_height;
}

/* grid_config.cpp:24-43 with BBox::contains (src/core/types.cpp:41-43):
 * inclusive bounds test, floor of a true division, then clamp (quirk Q1). */
int pcro_world_to_cell(const pcro_grid* g, double wx, double wy, int32_t* col, int32_t* row) {
    if (!(wx >= g->min_x && wx <= g->max_x && wy >= g->min_y && wy <= g->max_y)) return 0;
    int c = (int)floor((wx - g->min_x) / g->cell_size_x);
    int r = (int)floor((wy - g->max_y) / g->cell_size_y);
    if (c > g->width - 1) c = g->width - 1;
    if (c < 0) c = 0;
    if (r > g->height - 1) r = g->height - 1;
    if (r < 0) r = 0;
    *col = c;
    *row = r;
    return 1;
}

/* grid_config.cpp:81-91 */
void pcro_tile_cell_range(const pcro_grid* g, int32_t tile_row, int32_t tile_col,
                          int32_t* col_start, int32_t* row_start,
                          int32_t* col_count, int32_t* row_count) {
    *col_start = tile_col * g->tile_width;
    *row_start = tile_row * g->tile_height;
    int cw = g->width - *col_start, ch = g->height - *row_start;
    *col_count = g->tile_width < cw ? g->tile_width : cw;
    *row_count = g->tile_height < ch ? g->tile_height : ch;
}

/* ------------------------------------------------------------------------- */
/* Op algebra (include/pcr/ops/builtin_ops.h, src/ops/reduction_registry.cpp) */
/* ------------------------------------------------------------------------- */

/* reduction_registry.cpp:173-184: only these six are registered. */
int pcro_state_floats(int rtype) {
    switch (rtype) {
        case PCRO_SUM: case PCRO_MAX: case PCRO_MIN: case PCRO_COUNT: return 1;
        case PCRO_AVERAGE: case PCRO_WEIGHTED_AVERAGE: return 2;
        default: return 0;
    }
}

/* identity(): builtin_ops.h:13,26,39,52,65,82 ; init_state_cpu: registry.cpp:29-41 */
int pcro_init_state(int rtype, float* state, int64_t n) {
    int k = pcro_state_floats(rtype);
    if (!k) return fail(PCRO_INVALID_ARGUMENT, "unknown reduction type");
    float id = 0.0f;
    if (rtype == PCRO_MAX) id = -FLT_MAX;
    if (rtype == PCRO_MIN) id = FLT_MAX;
    for (int64_t i = 0; i < (int64_t)k * n; ++i) state[i] = id;
    return PCRO_OK;
}

/* accumulate_cpu<Op>: registry.cpp:50-110 (single-thread branch :93-106);
 * combine(): builtin_ops.h:14,27,40,53,66,86-88 (WeightedAverage combine == weight 1, quirk Q8). */
int pcro_accumulate(int rtype, const uint32_t* ci, const float* v, float* s,
                    size_t np, int64_t n) {
    if (!pcro_state_floats(rtype)) return fail(PCRO_INVALID_ARGUMENT, "unknown reduction type");
    for (size_t j = 0; j < np; ++j) {
        uint32_t c = ci[j];
        if (c >= (uint32_t)n) return fail(PCRO_INVALID_ARGUMENT, "cell index out of range");
        float val = v[j];
        switch (rtype) {
            case PCRO_SUM: s[c] = s[c] + val; break;
            case PCRO_MAX: s[c] = fmaxf(s[c], val); break;
            case PCRO_MIN: s[c] = fminf(s[c], val); break;
            case PCRO_COUNT: s[c] = s[c] + 1.0f; break;
            case PCRO_AVERAGE:
            case PCRO_WEIGHTED_AVERAGE:
                s[c] = s[c] + val;
                s[n + c] = s[n + c] + 1.0f;
                break;
        }
    }
    return PCRO_OK;
}

/* merge_state_cpu<Op>: registry.cpp:115-133 ; merge(): builtin_ops.h:15,28,41,54,67,95-97 */
int pcro_merge_state(int rtype, float* d, const float* s, int64_t n) {
    int k = pcro_state_floats(rtype);
    if (!k) return fail(PCRO_INVALID_ARGUMENT, "unknown reduction type");
    for (int64_t i = 0; i < (int64_t)k * n; ++i) {
        if (rtype == PCRO_MAX) d[i] = fmaxf(d[i], s[i]);
        else if (rtype == PCRO_MIN) d[i] = fminf(d[i], s[i]);
        else d[i] = d[i] + s[i];
    }
    return PCRO_OK;
}

/* finalize_cpu<Op>: registry.cpp:138-154 ; finalize(): builtin_ops.h:16,29,42,55,68-70,99-101
 * (Sum of an empty cell is 0.0, everything else NaN: quirk Q2). */
int pcro_finalize_state(int rtype, const float* s, float* out, int64_t n) {
    if (!pcro_state_floats(rtype)) return fail(PCRO_INVALID_ARGUMENT, "unknown reduction type");
    for (int64_t i = 0; i < n; ++i) {
        switch (rtype) {
            case PCRO_SUM: out[i] = s[i]; break;
            case PCRO_MAX: out[i] = (s[i] == -FLT_MAX) ? NAN : s[i]; break;
            case PCRO_MIN: out[i] = (s[i] == FLT_MAX) ? NAN : s[i]; break;
            case PCRO_COUNT: out[i] = (s[i] > 0.0f) ? s[i] : NAN; break;
            case PCRO_AVERAGE:
            case PCRO_WEIGHTED_AVERAGE:
                out[i] = (s[n + i] > 0.0f) ? s[i] / s[n + i] : NAN;
                break;
        }
    }
    return PCRO_OK;
}

/* ------------------------------------------------------------------------- */
/* Glyph splat (src/engine/glyph_kernels.cu, CPU half)                        */
/* ------------------------------------------------------------------------- */

/* update_state_cpu: glyph_kernels.cu:36-74.  `wide` keeps double state and adds the
 * exact product (test-only error bound, see header). */
static inline void upd(void* state, int wide, int64_t n, int64_t cell,
                       float val, float w, int rtype) {
    if (!wide) {
        float* s = (float*)state;
        switch (rtype) {
            case PCRO_WEIGHTED_AVERAGE:
            case PCRO_AVERAGE:
                s[cell] += val * w;
                s[n + cell] += w;
                break;
            case PCRO_SUM: s[cell] += val * w; break;
            case PCRO_COUNT: s[cell] += w; break;
            default: break;
        }
    } else {
        double* s = (double*)state;
        switch (rtype) {
            case PCRO_WEIGHTED_AVERAGE:
            case PCRO_AVERAGE:
                s[cell] += (double)val * (double)w;
                s[n + cell] += (double)w;
                break;
            case PCRO_SUM: s[cell] += (double)val * (double)w; break;
            case PCRO_COUNT: s[cell] += (double)w; break;
            default: break;
        }
    }
}

/* accumulate_glyph_gaussian_cpu: glyph_kernels.cu:79-183 (quirks Q4, Q5, Q6). */
static void gaussian_one(const pcro_glyph* spec, int rtype, const pcro_points* b, uint64_t p,
                         void* state, int wide, int64_t tile_cells, const pcro_grid* g,
                         int tc0, int tr0, int tw, int th) {
    const double inv_csx = 1.0 / g->cell_size_x;
    const double inv_csy = 1.0 / g->cell_size_y;
    float val = b->value[p];
    double fcx = (b->x[p] - g->min_x) * inv_csx;      /* :110 */
    double fcy = (b->y[p] - g->max_y) * inv_csy;      /* :111 */
    float sub_cx = (float)(fcx - floor(fcx));         /* :114 */
    float sub_cy = (float)(fcy - floor(fcy));

    float sx_world = (b->sigma_x && b->sigma_x[p] > 0.0f) ? b->sigma_x[p] : spec->default_sigma_x;
    float sy_world = (b->sigma_y && b->sigma_y[p] > 0.0f) ? b->sigma_y[p] : spec->default_sigma_y;
    float sx = sx_world * (float)inv_csx;             /* :126 */
    float sy = sy_world * (float)inv_csy;             /* negative on north-up grids */

    float rot = b->rotation ? b->rotation[p] : spec->default_rotation;
    float cos_rot = cosf(-rot);
    float sin_rot = sinf(-rot);

    float mx = sx > sy ? sx : sy;                     /* std::max(sx, sy) :134 */
    float R = 3.0f * mx;
    if (spec->max_radius_cells < R) R = spec->max_radius_cells;
    int r = (int)ceilf(R);

    int icx = (int)floor(fcx);
    int icy = (int)floor(fcy);

    for (int dy = -r; dy <= r; ++dy) {
        for (int dx = -r; dx <= r; ++dx) {
            int lx = icx + dx - tc0;
            int ly = icy + dy - tr0;
            if (lx < 0 || lx >= tw) continue;         /* tile clip :151-154 */
            if (ly < 0 || ly >= th) continue;
            float rdx = (float)dx - sub_cx;           /* corner sampling :157-158 */
            float rdy = (float)dy - sub_cy;
            float rxr = rdx * cos_rot + rdy * (-sin_rot);
            float ryr = rdx * sin_rot + rdy * cos_rot;
            float w = expf(-0.5f * ((rxr / sx) * (rxr / sx) + (ryr / sy) * (ryr / sy)));
            if (w < 1e-6f) continue;                  /* :166 */
            upd(state, wide, tile_cells, (int64_t)ly * tw + lx, val, w, rtype);
        }
    }
}

/* accumulate_glyph_line_cpu: glyph_kernels.cu:188-281 (quirk Q7). */
static void line_one(const pcro_glyph* spec, int rtype, const pcro_points* b, uint64_t p,
                     void* state, int wide, int64_t tile_cells, const pcro_grid* g,
                     int tc0, int tr0, int tw, int th) {
    const double inv_csx = 1.0 / g->cell_size_x;
    const double inv_csy = 1.0 / g->cell_size_y;
    float val = b->value[p];
    double fcx = (b->x[p] - g->min_x) * inv_csx;
    double fcy = (b->y[p] - g->max_y) * inv_csy;

    float direction = b->direction ? b->direction[p] : spec->default_direction;
    float half_len = b->half_length ? b->half_length[p] : spec->default_half_length;

    float hx = half_len * (float)inv_csx;             /* :228-229 */
    float hy = half_len * (float)inv_csy;
    float cap = spec->max_radius_cells;
    if (cap < hx) hx = cap;                           /* std::min(h, cap) :233-234 */
    if (cap < hy) hy = cap;

    float cos_d = cosf(direction);
    float sin_d = sinf(direction);

    double x0 = fcx - hx * cos_d;                     /* float product, double sum :240-243 */
    double y0 = fcy - hy * sin_d;
    double x1 = fcx + hx * cos_d;
    double y1 = fcy + hy * sin_d;

    int ix0 = (int)round(x0), iy0 = (int)round(y0);
    int ix1 = (int)round(x1), iy1 = (int)round(y1);

    int ddx = abs(ix1 - ix0), ddy = abs(iy1 - iy0);
    int sxs = ix0 < ix1 ? 1 : -1, sys = iy0 < iy1 ? 1 : -1;
    int err = ddx - ddy, cx = ix0, cy = iy0;
    int max_steps = 2 * (ddx + ddy) + 2;
    for (int step = 0; step <= max_steps; ++step) {
        int lx = cx - tc0, ly = cy - tr0;
        if (lx >= 0 && lx < tw && ly >= 0 && ly < th)
            upd(state, wide, tile_cells, (int64_t)ly * tw + lx, val, 1.0f, rtype);
        if (cx == ix1 && cy == iy1) break;
        int e2 = 2 * err;
        if (e2 > -ddy) { err -= ddy; cx += sxs; }
        if (e2 < ddx) { err += ddx; cy += sys; }
    }
}

static int glyph_rtype_ok(int rtype) {
    return rtype == PCRO_WEIGHTED_AVERAGE || rtype == PCRO_AVERAGE ||
           rtype == PCRO_SUM || rtype == PCRO_COUNT;
}

static int glyph_batch(const pcro_glyph* spec, int rtype, const pcro_points* b,
                       void* state, int wide, int64_t tile_cells, const pcro_grid* g,
                       int tc0, int tr0, int tw, int th) {
    if (b->n == 0) return PCRO_OK;                    /* :584 */
    /* accumulate_glyph_cpu: :286-326 */
    if (!glyph_rtype_ok(rtype))
        return fail(PCRO_NOT_IMPLEMENTED,
                    "glyph splatting only supports WeightedAverage, Average, Sum, or Count reduction types");
    if (spec->type == PCRO_GLYPH_POINT)
        return fail(PCRO_INVALID_ARGUMENT, "accumulate_glyph: Point glyph should use regular accumulate()");
    if (!b->x || !b->y)
        return fail(PCRO_INVALID_ARGUMENT, "glyph: batch.coord_x / coord_y must be non-null");
    for (uint64_t p = 0; p < b->n; ++p) {
        if (spec->type == PCRO_GLYPH_GAUSSIAN)
            gaussian_one(spec, rtype, b, p, state, wide, tile_cells, g, tc0, tr0, tw, th);
        else if (spec->type == PCRO_GLYPH_LINE)
            line_one(spec, rtype, b, p, state, wide, tile_cells, g, tc0, tr0, tw, th);
        else
            return fail(PCRO_NOT_IMPLEMENTED, "glyph: unknown glyph type");
    }
    return PCRO_OK;
}

int pcro_accumulate_glyph(const pcro_glyph* spec, int rtype, const pcro_points* batch,
                          float* state, int64_t tile_cells, const pcro_grid* g,
                          int32_t tc0, int32_t tr0, int32_t tw, int32_t th) {
    return glyph_batch(spec, rtype, batch, state, 0, tile_cells, g, tc0, tr0, tw, th);
}

/* ------------------------------------------------------------------------- */
/* Point filter (src/engine/filter.cpp)                                        */
/* ------------------------------------------------------------------------- */

/* evaluate_predicate: filter.cpp:34-56 */
static int eval_pred(const pcro_predicate* p, float v) {
    int in = 0;
    switch (p->op) {
        case PCRO_CMP_EQUAL: return v == p->value;
        case PCRO_CMP_NOT_EQUAL: return v != p->value;
        case PCRO_CMP_LESS: return v < p->value;
        case PCRO_CMP_LESS_EQUAL: return v <= p->value;
        case PCRO_CMP_GREATER: return v > p->value;
        case PCRO_CMP_GREATER_EQUAL: return v >= p->value;
        case PCRO_CMP_IN_SET:
        case PCRO_CMP_NOT_IN_SET:
            for (int k = 0; k < p->set_size; ++k) in = in || (v == p->set[k]);   /* std::find on floats */
            return p->op == PCRO_CMP_IN_SET ? in : !in;
    }
    return 0;
}

/* the AND loop of filter_points_cpu: filter.cpp:141-206 (a point passes iff every predicate holds) */
uint64_t pcro_filter_mask(const pcro_predicate* preds, int n_pred, uint64_t n, uint8_t* mask) {
    uint64_t kept = 0;
    for (uint64_t i = 0; i < n; ++i) {
        int ok = 1;
        for (int k = 0; k < n_pred && ok; ++k) ok = eval_pred(&preds[k], preds[k].channel[i]);
        mask[i] = (uint8_t)ok;
        kept += (uint64_t)ok;
    }
    return kept;
}

/* ------------------------------------------------------------------------- */
/* Whole path for one ReductionSpec (src/engine/pipeline.cpp)                 */
/* ------------------------------------------------------------------------- */

struct pcro_reduction {
    pcro_grid g;
    pcro_glyph glyph;
    int rtype;
    int wide;
    int k;
    int tiles_x, tiles_y;
    void** tile_state;      /* per tile, NULL until first touched; K * tile_cells */
    uint64_t points_valid;
};

pcro_reduction* pcro_create(const pcro_grid* g, int rtype, const pcro_glyph* glyph, int wide) {
    int k = pcro_state_floats(rtype);
    if (!k) { fail(PCRO_INVALID_ARGUMENT, "pipeline: unknown reduction type"); return NULL; } /* pipeline.cpp:229-233 */
    if (g->width <= 0 || g->height <= 0 || g->tile_width <= 0 || g->tile_height <= 0) {
        fail(PCRO_INVALID_ARGUMENT, "pipeline: grid/tile dimensions must be positive");
        return NULL;
    }
    pcro_reduction* r = (pcro_reduction*)calloc(1, sizeof *r);
    if (!r) return NULL;
    r->g = *g;
    if (glyph) r->glyph = *glyph; else r->glyph.type = PCRO_GLYPH_POINT;
    r->rtype = rtype;
    r->wide = wide;
    r->k = k;
    r->tiles_x = (g->width + g->tile_width - 1) / g->tile_width;
    r->tiles_y = (g->height + g->tile_height - 1) / g->tile_height;
    r->tile_state = (void**)calloc((size_t)r->tiles_x * r->tiles_y, sizeof(void*));
    if (!r->tile_state) { free(r); return NULL; }
    return r;
}

void pcro_destroy(pcro_reduction* r) {
    if (!r) return;
    for (int i = 0; i < r->tiles_x * r->tiles_y; ++i) free(r->tile_state[i]);
    free(r->tile_state);
    free(r);
}

/* TileManager::acquire on a miss = identity init (tile_manager.cpp:183-375). */
static void* acquire_tile(pcro_reduction* r, int trow, int tcol, int64_t* cells_out,
                          int* c0, int* r0, int* cw, int* ch) {
    pcro_tile_cell_range(&r->g, trow, tcol, c0, r0, cw, ch);
    int64_t cells = (int64_t)(*cw) * (*ch);
    *cells_out = cells;
    void** slot = &r->tile_state[trow * r->tiles_x + tcol];
    if (!*slot) {
        size_t n = (size_t)r->k * (size_t)cells;
        if (r->wide) {
            double* d = (double*)malloc(n * sizeof(double));
            if (!d) return NULL;
            double id = r->rtype == PCRO_MAX ? -(double)FLT_MAX : r->rtype == PCRO_MIN ? (double)FLT_MAX : 0.0;
            for (size_t i = 0; i < n; ++i) d[i] = id;
            *slot = d;
        } else {
            float* f = (float*)malloc(n * sizeof(float));
            if (!f) return NULL;
            pcro_init_state(r->rtype, f, cells);
            *slot = f;
        }
    }
    return *slot;
}

/* process_cloud for one reduction: pipeline.cpp:360-742, with assign (tile_router.cpp:51-126)
 * and the tile-local index (tile_router.cpp:253-366) fused per point; no sort. */
int pcro_ingest(pcro_reduction* r, const pcro_points* pts) {
    if (pts->n == 0) return PCRO_OK;                  /* pipeline.cpp:284-287 */
    const int use_glyph = r->glyph.type != PCRO_GLYPH_POINT;
    if (use_glyph && !glyph_rtype_ok(r->rtype))       /* pipeline.cpp:500-508 */
        return fail(PCRO_NOT_IMPLEMENTED,
                    "pipeline: glyph splatting only supports WeightedAverage, Average, Sum, or Count reduction types");
    for (uint64_t p = 0; p < pts->n; ++p) {
        int32_t col, row;
        if (!pcro_world_to_cell(&r->g, pts->x[p], pts->y[p], &col, &row)) continue;
        r->points_valid++;
        int tcol = col / r->g.tile_width, trow = row / r->g.tile_height;
        int c0, r0, cw, ch;
        int64_t cells;
        void* st = acquire_tile(r, trow, tcol, &cells, &c0, &r0, &cw, &ch);
        if (!st) return fail(PCRO_OUT_OF_MEMORY, "oracle: out of memory");
        if (use_glyph) {
            pcro_points one = *pts;
            one.n = 1;
            /* glyph functions index with p, so hand them the arrays and call per point */
            if (r->glyph.type == PCRO_GLYPH_GAUSSIAN)
                gaussian_one(&r->glyph, r->rtype, pts, p, st, r->wide, cells, &r->g, c0, r0, cw, ch);
            else if (r->glyph.type == PCRO_GLYPH_LINE)
                line_one(&r->glyph, r->rtype, pts, p, st, r->wide, cells, &r->g, c0, r0, cw, ch);
            else
                return fail(PCRO_NOT_IMPLEMENTED, "glyph: unknown glyph type");
            (void)one;
        } else {
            /* local index uses the ACTUAL (edge-clamped) tile width: tile_router.cpp:253-366 */
            uint32_t lc = (uint32_t)((row - r0) * cw + (col - c0));
            float v = pts->value[p];
            if (!r->wide) {
                int rc = pcro_accumulate(r->rtype, &lc, &v, (float*)st, 1, cells);
                if (rc) return rc;
            } else {
                double* s = (double*)st;
                switch (r->rtype) {
                    case PCRO_SUM: s[lc] += (double)v; break;
                    case PCRO_MAX: s[lc] = fmax(s[lc], (double)v); break;
                    case PCRO_MIN: s[lc] = fmin(s[lc], (double)v); break;
                    case PCRO_COUNT: s[lc] += 1.0; break;
                    default: s[lc] += (double)v; s[cells + lc] += 1.0; break;
                }
            }
        }
    }
    return PCRO_OK;
}

/* finalize_result: pipeline.cpp:1154-1286.  NaN-fill, skip tiles without state (Q3),
 * finalize each tile, copy its block into the row-major band. */
int pcro_finalize(const pcro_reduction* r, float* band) {
    const pcro_grid* g = &r->g;
    int64_t total = (int64_t)g->width * g->height;
    for (int64_t i = 0; i < total; ++i) band[i] = NAN;
    for (int ty = 0; ty < r->tiles_y; ++ty) {
        for (int tx = 0; tx < r->tiles_x; ++tx) {
            void* st = r->tile_state[ty * r->tiles_x + tx];
            if (!st) continue;
            int c0, r0, cw, ch;
            pcro_tile_cell_range(g, ty, tx, &c0, &r0, &cw, &ch);
            int64_t cells = (int64_t)cw * ch;
            float* fin = (float*)malloc((size_t)cells * sizeof(float));
            if (!fin) return fail(PCRO_OUT_OF_MEMORY, "oracle: out of memory");
            if (!r->wide) {
                pcro_finalize_state(r->rtype, (const float*)st, fin, cells);
            } else {
                const double* s = (const double*)st;
                for (int64_t i = 0; i < cells; ++i) {
                    switch (r->rtype) {
                        case PCRO_SUM: fin[i] = (float)s[i]; break;
                        case PCRO_MAX: fin[i] = (s[i] == -(double)FLT_MAX) ? NAN : (float)s[i]; break;
                        case PCRO_MIN: fin[i] = (s[i] == (double)FLT_MAX) ? NAN : (float)s[i]; break;
                        case PCRO_COUNT: fin[i] = s[i] > 0.0 ? (float)s[i] : NAN; break;
                        default: fin[i] = s[cells + i] > 0.0 ? (float)(s[i] / s[cells + i]) : NAN; break;
                    }
                }
            }
            for (int ly = 0; ly < ch; ++ly)
                memcpy(band + (int64_t)(r0 + ly) * g->width + c0, fin + (int64_t)ly * cw,
                       (size_t)cw * sizeof(float));
            free(fin);
        }
    }
    return PCRO_OK;
}

int pcro_tile_touched(const pcro_reduction* r, int32_t trow, int32_t tcol) {
    if (trow < 0 || trow >= r->tiles_y || tcol < 0 || tcol >= r->tiles_x) return 0;
    return r->tile_state[trow * r->tiles_x + tcol] != NULL;
}

uint64_t pcro_points_valid(const pcro_reduction* r) { return r->points_valid; }
