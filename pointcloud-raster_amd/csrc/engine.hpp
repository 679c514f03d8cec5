// engine.hpp -- the scatter engine object behind pcr_hip_engine_* / pcr_hip_scatter_*.
#pragma once

#include "common.hpp"

#include <map>
#include <vector>

struct pcr_hip_engine {
    pcr_hip_grid grid{};
    pcrhip::GridDev gd{};
    hipStream_t stream = nullptr;
    int device = 0;
    int num_cus = 256;

    uint32_t* d_touched = nullptr;             // tiles_x * tiles_y words
    int ntiles = 0;
    unsigned long long* d_counters = nullptr;  // [0] = valid points of the last scatter

    int forced_path = 0;                       // 0 auto, 1 direct, 2 binned, 3 moments (Gaussian only)
    int max_bins = 0;                          // LDS tiles per binning pass (kMaxBins; PCR_HIP_DEBUG_MAX_BINS lowers it
                                               // so that tests reach the large-grid paths on small grids)
    int stats_scatter_chunk = 0;               // points per k_bin_scatter workgroup of the last binned scatter
    bool two_level = true;                     // PCR_HIP_DEBUG_TWO_LEVEL=0 forces the row-band sweep instead
    pcr_hip_scatter_stats stats{};
    int planes_fresh = 0;                      // pcr_hip_engine_planes_fresh, for the NEXT scatter: 0 its planes hold earlier
                                               // contributions, 1 they hold identity values, 2 they are UNDEFINED (the
                                               // scatter defines every cell of the state window, see engine.hip)

    // pcr_hip_engine_finalize_with_scatter, for the NEXT Point scatter: bands its tile pass may store (scatter_binned.hip)
    pcrhip::FinalizeOuts fused_outs{};
    uint32_t* fused_done = nullptr;
    bool fused_taken = false;

    // optional per-kernel event timing
    bool profiling = false;
    std::string profile_only;                  // non-empty: only launches timed under this name are bracketed by events
    struct Pending { const char* name; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::map<std::string, std::pair<uint32_t, double>> kernel_ms;

    // scratch of the binned / moment paths: borrowed per scatter from the device-wide arena (engine.hip)
    char* d_scratch = nullptr;
    size_t scratch_cap = 0;
    bool scratch_borrowed = false;
};

namespace pcrhip {

int ensure_scratch(pcr_hip_engine* e, size_t bytes);
void release_scratch(pcr_hip_engine* e);
// Device-resident tap tables of the moment path, shared by the engines of a device (engine.hip); call with the scratch
// borrowed.  fill(tables, K, r, sx, sy) builds the host copy when the glyph spec changed.
int shared_taps(pcr_hip_engine* e, int K, int r, float sx, float sy,
                void (*fill)(std::vector<float>&, int, int, float, float), const float** d_taps, size_t* count);

// Brackets one kernel launch with events when profiling is on.
struct ScopedKernelTimer {
    pcr_hip_engine* e;
    const char* name;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedKernelTimer(pcr_hip_engine* eng, const char* nm) : e(eng), name(nm) {
        if (!e->profiling) return;
        if (!e->profile_only.empty() && e->profile_only != nm) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, e->stream);
    }
    ~ScopedKernelTimer() {
        if (!a) return;
        (void)hipEventRecord(b, e->stream);
        e->pending.push_back({name, a, b});
    }
};

// ---- binning (counting sort of points by LDS tile), scatter_binned.hip ------------------------
struct BinGeom {
    int tile_w, tile_h;                 // interior of an LDS tile, cells (a bin owns these cells)
    int bins_x, bins_y, nbins;
    int chunk;                          // points per workgroup in the count / scatter passes
    int row0, rows;                     // the band of state rows [row0, row0 + rows) the bins cover (window-relative)
    int sup_shift;                      // two-level sort: the first level groups 2^sup_shift consecutive tiles (0: one level)
};
constexpr int kMaxBands = 32;           // a grid with more LDS tiles than kMaxBins is swept in row bands
struct BinItem {                        // one workgroup's share of a bin's records
    unsigned bin, first, count, shared; // shared != 0: the bin was split, merge with atomics
};
struct BinBuffers {                     // device pointers into the engine's scratch arena
    const uint2* records;               // Value / Index records, grouped by bin; .x = local cell
    const BinItem* items;
    const unsigned* n_items;            // [0] = items, [1] = 1 when some bin was split into several items
    int max_items;
};
constexpr int kMaxBins = 12160;        // scatter pass LDS: the 8192-record staging window (64 KB) + 8 B per bin = 159 KB of the CU's 160.
                                        // Round 5 (8064 until then, "beyond this a block's runs are single records anyway"): measured on
                                        // the window of a C5 shard at N = 2 (16384 x 8192, 11 008 tiles, 500 M points) one level with
                                        // runs of ~2 records costs 5.79 ms a step, the two-level sort 7.18, two row bands 9.09
                                        // (tools/n2_shard_ab.sh) -- the second level's two passes over the records cost more than the
                                        // partial lines do.  Beyond this: two-level sort / row bands.
constexpr int kLcellBits = 15;          // up to 32768 cells per LDS tile
constexpr int kMaxTiles = 1 << (32 - kLcellBits);   // routing key = tile << 15 | local cell
constexpr int kMaxSubBins = 2048;       // tiles per first-level group (second-level scatter: 128 KB staging + 12 B per tile)

// Rows per band so that a band's bins fit the binning passes (whole tile rows); 0 = cannot be banded.
inline int band_rows_for(const GridDev& g, int tile_w, int tile_h, int max_bins) {
    const int bins_x = (g.W + tile_w - 1) / tile_w;
    const int bins_y = max_bins / bins_x;
    if (bins_y < 1) return 0;
    const int64_t rows = (int64_t)bins_y * tile_h;
    return (int)(rows < g.st_rows ? rows : g.st_rows);
}

// Passes A (histogram + routing keys), scan, B (LDS-staged scatter) over the points that gd owns (for a band:
// the engine's grid with the owned rows narrowed to the band).  Record kinds:
//   Value  8 B {local cell, value}         Point glyph
//   Index  8 B {local cell, point index}   Gaussian tiles with per-point sigma / rotation channels or r > 3 (the others, and
//                                          Lines, bin 16-byte value records: bin16.hpp)
enum class RecordKind { Value, Index };
// every_bin: an item (possibly of zero records) for EVERY bin, so that the tile pass visits every cell of the band.
int bin_points(pcr_hip_engine* e, const GridDev& gd, const BinGeom& b, const double* x, const double* y, const float* v,
               uint64_t n, RecordKind kind, const GlyphDev* gl, unsigned item_records, BinBuffers* out, bool every_bin = false);
// Identity values (0, 0, -FLT_MAX, +FLT_MAX) into the planes of `mask` over the engine's state window (engine.hip).
int fill_identity(pcr_hip_engine* e, uint32_t mask, const PlanesDev& pl);

// Two-level counting sort (groups of 2^shift tiles, then tiles) for windows with more tiles than one pass counts;
// 8-byte Value or Index records.  two_level_shift: 0 when not applicable (disabled, or more than kMaxTiles tiles).
int two_level_shift(const pcr_hip_engine* e, int tiles);
int bin_points_two_level(pcr_hip_engine* e, const BinGeom& tiles, const double* x, const double* y, const float* v,
                         uint64_t n, bool index_records, unsigned item_records, BinBuffers* out, bool every_bin = false);

// direct path (global atomics), scatter_direct.hip
int direct_point(pcr_hip_engine* e, uint32_t mask, const PlanesDev& pl,
                 const double* x, const double* y, const float* v, uint64_t n);
int direct_glyph(pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask, const PlanesDev& pl,
                 const double* x, const double* y, const float* v, uint64_t n);

// binned path (LDS tiles), scatter_binned.hip
bool binned_point_supported(const pcr_hip_engine* e, uint32_t mask);
int binned_point(pcr_hip_engine* e, uint32_t mask, const PlanesDev& pl,
                 const double* x, const double* y, const float* v, uint64_t n);
bool binned_glyph_supported(const pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask);
int binned_glyph(pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask, const PlanesDev& pl,
                 const double* x, const double* y, const float* v, uint64_t n);

// Gaussian cell tiles on 16-byte value records (default-sigma, unrotated, r <= 3), scatter_cells.hip
bool cells_gauss_supported(const pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask);
int cells_gauss(pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask, const PlanesDev& pl,
                const double* x, const double* y, const float* v, uint64_t n);

// separable moment + convolution path for large default-sigma Gaussians, scatter_moments.hip
bool moments_supported(const pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask);
// planes_undefined: the planes were only allocated (pcr_hip_engine_planes_fresh(e, 2)); the path defines every cell itself
int moments_gauss(pcr_hip_engine* e, const GlyphDev& gl, uint32_t mask, const PlanesDev& pl,
                  const double* x, const double* y, const float* v, uint64_t n, bool planes_undefined);

}  // namespace pcrhip
