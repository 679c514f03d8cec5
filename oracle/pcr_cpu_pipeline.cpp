/*
 * pcr_cpu_pipeline.cpp -- the reference's CPU engine restated STAGE BY STAGE, for timing.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/pcr_oracle.h).  bench.py's `cpu_baseline` leg times it on the
 * GPU box's host cores next to the HIP engine; tests/test_cpu_pipeline.py checks that it returns what
 * the oracle returns.  Nothing under pointcloud-raster_amd/ links or calls it.
 *
 * pcr_oracle.c folds points into the grid in input order, which is the reference's ARITHMETIC but not
 * its COST: the reference's CPU path spends ~70 % of its time in a serial std::sort and takes a lock
 * per update.  This file keeps those stages, per ReductionSpec, exactly as process_cloud runs them
 * (src/engine/pipeline.cpp:283-770):
 *
 *   assign           world_to_cell per point, `omp parallel for`        src/engine/tile_router.cpp:84-123
 *   sort             index vector of size_t, std::sort by (valid, tile, cell), SERIAL,
 *                    then gather of cells / tiles / valid / values (+ glyph arrays)   tile_router.cpp:138-240
 *   extract_batches  one batch per run of equal tile, local cell indices    tile_router.cpp:242-366
 *   accumulate       Point glyph: `omp parallel for` over the batch with the read-modify-write of
 *                    EVERY point inside `omp critical`                        src/ops/reduction_registry.cpp:63-92
 *                    Line / Gaussian: pcro_accumulate_glyph, single thread     src/engine/glyph_kernels.cu:79-281
 *   finalize         Op::finalize per touched tile into a NaN-filled band    pipeline.cpp:1204-1286
 *
 * Not restated: the tile manager's LRU and its `.pcrt` flush on finalize (disk I/O, excluded from the
 * baseline on purpose -- it only makes the reference slower), the H2D/D2H copies of its GPU mode.
 * The op algebra, world_to_cell and the glyph splat are pcr_oracle.c's (pinned against the reference
 * itself, tests/test_oracle_vs_ref.py); this file only adds the reference's control flow around them.
 */
#include "pcr_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// builtin_ops.h restated as (state floats, combine) on a band-sequential tile state
inline void combine(int rtype, float* state, uint32_t cell, int64_t tile_cells, float v) {
    switch (rtype) {
        case PCRO_SUM: state[cell] += v; break;
        case PCRO_MAX: state[cell] = std::fmax(state[cell], v); break;
        case PCRO_MIN: state[cell] = std::fmin(state[cell], v); break;
        case PCRO_COUNT: state[cell] += 1.0f; break;
        default:                                   // Average {sum, count}; WeightedAverage {wsum, wgt} with w = 1
            state[cell] += v;
            state[tile_cells + cell] += 1.0f;
            break;
    }
}

}  // namespace

extern "C" {

/* Runs ONE ReductionSpec over one cloud the way the reference's CPU engine does and writes the finalized
 * band (W*H floats, NaN where the tile has no state).  threads <= 0: OpenMP default.
 * stage_seconds[5] (optional): assign, sort, extract_batches, accumulate, finalize.
 * Returns a PCRO_* status. */
int pcro_cpu_pipeline_run(const pcro_grid* g, int rtype, const pcro_glyph* glyph, const pcro_points* pts,
                          int threads, float* band, double* stage_seconds) {
    if (!g || !pts || !band) return PCRO_INVALID_ARGUMENT;
    const int K = pcro_state_floats(rtype);
    if (K <= 0) return PCRO_INVALID_ARGUMENT;
    const bool use_glyph = glyph && glyph->type != PCRO_GLYPH_POINT;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    const size_t n = pts->n;
    const int W = g->width, H = g->height;
    const int tiles_x = (W + g->tile_width - 1) / g->tile_width;
    const int tiles_y = (H + g->tile_height - 1) / g->tile_height;
    double t[6];
    t[0] = now_s();

    // ---- assign (tile_router.cpp:84-123) ----
    std::vector<uint32_t> cell(n), tile(n);
    std::vector<uint8_t> valid(n);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        int32_t cx, cy;
        if (!pcro_world_to_cell(g, pts->x[i], pts->y[i], &cx, &cy)) {
            valid[i] = 0; cell[i] = 0; tile[i] = 0;
            continue;
        }
        valid[i] = 1;
        cell[i] = (uint32_t)(cy * W + cx);
        tile[i] = (uint32_t)((cy / g->tile_height) * tiles_x + cx / g->tile_width);
    }
    t[1] = now_s();

    // ---- sort (tile_router.cpp:138-240): serial std::sort of an index vector, then gathers ----
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        if (!valid[a]) return false;
        if (!valid[b]) return true;
        if (tile[a] != tile[b]) return tile[a] < tile[b];
        return cell[a] < cell[b];
    });
    std::vector<uint32_t> s_cell(n), s_tile(n);
    std::vector<uint8_t> s_valid(n);
    std::vector<float> s_val(n);
    for (size_t i = 0; i < n; ++i) {
        const size_t src = order[i];
        s_cell[i] = cell[src]; s_tile[i] = tile[src]; s_valid[i] = valid[src];
    }
    for (size_t i = 0; i < n; ++i) s_val[i] = pts->value[order[i]];
    std::vector<double> s_x, s_y;
    std::vector<float> s_ch[5];
    const float* ch_src[5] = {pts->direction, pts->half_length, pts->sigma_x, pts->sigma_y, pts->rotation};
    if (use_glyph) {
        s_x.resize(n); s_y.resize(n);
        for (size_t i = 0; i < n; ++i) s_x[i] = pts->x[order[i]];
        for (size_t i = 0; i < n; ++i) s_y[i] = pts->y[order[i]];
        for (int c = 0; c < 5; ++c)
            if (ch_src[c]) {
                s_ch[c].resize(n);
                for (size_t i = 0; i < n; ++i) s_ch[c][i] = ch_src[c][order[i]];
            }
    }
    t[2] = now_s();

    // ---- extract_batches (tile_router.cpp:242-366): runs of equal tile, local cell indices ----
    struct Batch { uint32_t tile; size_t first, count; };
    std::vector<Batch> batches;
    std::vector<uint32_t> local(n);
    size_t n_valid = 0;
    while (n_valid < n && s_valid[n_valid]) ++n_valid;           // invalid points were sorted to the end
    for (size_t i = 0; i < n_valid;) {
        size_t j = i;
        while (j < n_valid && s_tile[j] == s_tile[i]) ++j;
        const int tr = (int)(s_tile[i] / (uint32_t)tiles_x), tc = (int)(s_tile[i] % (uint32_t)tiles_x);
        int32_t c0, r0, nc, nr;
        pcro_tile_cell_range(g, tr, tc, &c0, &r0, &nc, &nr);
        for (size_t k = i; k < j; ++k) {
            const int cy = (int)(s_cell[k] / (uint32_t)W), cx = (int)(s_cell[k] % (uint32_t)W);
            local[k] = (uint32_t)((cy - r0) * nc + (cx - c0));
        }
        batches.push_back({s_tile[i], i, j - i});
        i = j;
    }
    t[3] = now_s();

    // ---- accumulate per tile (pipeline.cpp:681-738) ----
    std::vector<std::vector<float>> state((size_t)tiles_x * tiles_y);
    for (const Batch& b : batches) {
        const int tr = (int)(b.tile / (uint32_t)tiles_x), tc = (int)(b.tile % (uint32_t)tiles_x);
        int32_t c0, r0, nc, nr;
        pcro_tile_cell_range(g, tr, tc, &c0, &r0, &nc, &nr);
        const int64_t tile_cells = (int64_t)nc * nr;
        std::vector<float>& st = state[b.tile];
        if (st.empty()) {                                          // tile manager: first acquire initialises
            st.resize((size_t)K * tile_cells);
            pcro_init_state(rtype, st.data(), tile_cells);
        }
        if (use_glyph) {
            pcro_points bp{};
            bp.x = s_x.data() + b.first; bp.y = s_y.data() + b.first; bp.value = s_val.data() + b.first;
            bp.direction = ch_src[0] ? s_ch[0].data() + b.first : nullptr;
            bp.half_length = ch_src[1] ? s_ch[1].data() + b.first : nullptr;
            bp.sigma_x = ch_src[2] ? s_ch[2].data() + b.first : nullptr;
            bp.sigma_y = ch_src[3] ? s_ch[3].data() + b.first : nullptr;
            bp.rotation = ch_src[4] ? s_ch[4].data() + b.first : nullptr;
            bp.n = b.count;
            int rc = pcro_accumulate_glyph(glyph, rtype, &bp, st.data(), tile_cells, g, c0, r0, nc, nr);
            if (rc) return rc;
        } else {
            // reduction_registry.cpp:63-92: every update under `omp critical`
            const uint32_t* lc = local.data() + b.first;
            const float* vv = s_val.data() + b.first;
            float* sp = st.data();
            const size_t m = b.count;
            bool bad = false;
#pragma omp parallel for schedule(static) shared(bad)
            for (size_t j = 0; j < m; ++j) {
                if (bad) continue;
                const uint32_t c = lc[j];
                if (c >= (uint32_t)tile_cells) {
#pragma omp critical
                    bad = true;
                    continue;
                }
                const float v = vv[j];
#pragma omp critical
                { combine(rtype, sp, c, tile_cells, v); }
            }
            if (bad) return PCRO_INVALID_ARGUMENT;
        }
    }
    t[4] = now_s();

    // ---- finalize (pipeline.cpp:1204-1286): NaN band, touched tiles finalized and block-copied ----
    const float nan = std::numeric_limits<float>::quiet_NaN();
    for (int64_t i = 0; i < (int64_t)W * H; ++i) band[i] = nan;
    std::vector<float> fin;
    for (int tr = 0; tr < tiles_y; ++tr)
        for (int tc = 0; tc < tiles_x; ++tc) {
            const std::vector<float>& st = state[(size_t)tr * tiles_x + tc];
            if (st.empty()) continue;
            int32_t c0, r0, nc, nr;
            pcro_tile_cell_range(g, tr, tc, &c0, &r0, &nc, &nr);
            fin.resize((size_t)nc * nr);
            pcro_finalize_state(rtype, st.data(), fin.data(), (int64_t)nc * nr);
            for (int y = 0; y < nr; ++y)
                std::memcpy(band + (size_t)(r0 + y) * W + c0, fin.data() + (size_t)y * nc, (size_t)nc * sizeof(float));
        }
    t[5] = now_s();
    if (stage_seconds)
        for (int k = 0; k < 5; ++k) stage_seconds[k] = t[k + 1] - t[k];
    return PCRO_OK;
}

int pcro_cpu_pipeline_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
