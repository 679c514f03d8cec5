// host_engine.cpp -- see host_engine.h.  Arithmetic after the reference's CPU path, cited at each step; structure ours.
#include "host_engine.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <limits>

#include <cstdio>

#include <omp.h>
#include <sched.h>

namespace pcr {
namespace detail {

namespace {

constexpr int kMaxLineExtent = 1 << 23;      // a longer segment is dropped, as by the HIP engine

// Inclusive row range [lo, hi] -> the stripes it meets.
struct Span { int lo, hi; };

// The reference tile of a routed cell as a clip rectangle [c0, c1) x [r0, r1) (glyph_kernels.cu:151-154, 263-267: a footprint
// only touches cells of the tile that holds the point's centre cell).
struct Clip { int c0, c1, r0, r1; };

inline Clip tile_clip(const GridConfig& g, int col, int row) {
    Clip k;
    k.c0 = (col / g.tile_width) * g.tile_width;
    k.r0 = (row / g.tile_height) * g.tile_height;
    k.c1 = std::min(k.c0 + g.tile_width, g.width);
    k.r1 = std::min(k.r0 + g.tile_height, g.height);
    return k;
}

// One Gaussian footprint, everything that does not depend on the cell (glyph_kernels.cu:101-140).
struct Splat {
    int icx, icy, r;
    float sub_cx, sub_cy, sx, sy, cos_rot, sin_rot;
    Clip clip;
    bool ok;
};

inline Splat make_splat(const GridConfig& g, const GlyphSpec& spec, const HostGlyphArrays& a, size_t i, double wx, double wy,
                        int col, int row) {
    Splat s{};
    const double inv_csx = 1.0 / g.cell_size_x, inv_csy = 1.0 / g.cell_size_y;
    const double fcx = (wx - g.bounds.min_x) * inv_csx;
    const double fcy = (wy - g.bounds.max_y) * inv_csy;
    s.sub_cx = static_cast<float>(fcx - std::floor(fcx));
    s.sub_cy = static_cast<float>(fcy - std::floor(fcy));
    const float sx_world = (a.sigma_x && a.sigma_x[i] > 0.0f) ? a.sigma_x[i] : spec.default_sigma_x;     // :120-123
    const float sy_world = (a.sigma_y && a.sigma_y[i] > 0.0f) ? a.sigma_y[i] : spec.default_sigma_y;
    s.sx = sx_world * static_cast<float>(inv_csx);
    s.sy = sy_world * static_cast<float>(inv_csy);                   // negative on north-up grids: only ever squared (Q6)
    const float rot = a.rotation ? a.rotation[i] : spec.default_rotation;
    s.cos_rot = std::cos(-rot);
    s.sin_rot = std::sin(-rot);
    const float R = std::min(3.0f * std::max(s.sx, s.sy), spec.max_radius_cells);
    s.ok = std::isfinite(R) && std::isfinite(fcx) && std::isfinite(fcy);
    if (!s.ok) return s;
    s.r = static_cast<int>(std::ceil(std::min(R, 1048576.0f)));
    s.icx = static_cast<int>(std::floor(fcx));
    s.icy = static_cast<int>(std::floor(fcy));
    s.clip = tile_clip(g, col, row);
    return s;
}

// One Line segment: rounded end points (glyph_kernels.cu:213-250; end points in double, `round`, Q7).
struct Segment {
    int ix0, iy0, ix1, iy1;
    Clip clip;
    bool ok;
};

inline Segment make_segment(const GridConfig& g, const GlyphSpec& spec, const HostGlyphArrays& a, size_t i, double wx, double wy,
                            int col, int row) {
    Segment s{};
    const double inv_csx = 1.0 / g.cell_size_x, inv_csy = 1.0 / g.cell_size_y;
    const double fcx = (wx - g.bounds.min_x) * inv_csx;
    const double fcy = (wy - g.bounds.max_y) * inv_csy;
    const float direction = a.direction ? a.direction[i] : spec.default_direction;
    const float half_len = a.half_length ? a.half_length[i] : spec.default_half_length;
    float hx = half_len * static_cast<float>(inv_csx);
    float hy = half_len * static_cast<float>(inv_csy);
    const float cap = spec.max_radius_cells;
    hx = std::min(hx, cap);
    hy = std::min(hy, cap);                                          // (negative hy passes untouched: never capped)
    const float cos_d = std::cos(direction), sin_d = std::sin(direction);
    const double x0 = fcx - hx * cos_d, y0 = fcy - hy * sin_d;
    const double x1 = fcx + hx * cos_d, y1 = fcy + hy * sin_d;
    const double lim = 1073741824.0;
    s.ok = std::fabs(x0) < lim && std::fabs(y0) < lim && std::fabs(x1) < lim && std::fabs(y1) < lim;     // (false for NaN)
    if (!s.ok) return s;
    s.ix0 = static_cast<int>(std::round(x0));
    s.iy0 = static_cast<int>(std::round(y0));
    s.ix1 = static_cast<int>(std::round(x1));
    s.iy1 = static_cast<int>(std::round(y1));
    s.ok = std::abs(s.ix1 - s.ix0) <= kMaxLineExtent && std::abs(s.iy1 - s.iy0) <= kMaxLineExtent;
    s.clip = tile_clip(g, col, row);
    return s;
}

}  // namespace

// cpu_threads = 0 means "every core" (include/pcr/engine/pipeline.h:78) -- every core this process may actually USE: the
// affinity mask and the cgroup's CPU quota (cpu.max) bound it.  A 256-thread team on a 16-CPU quota spends its time being
// throttled at barriers (measured on the GPU box: 6 Mpts/s with 256 threads where one thread does 25).
static int usable_cpus() {
    int n = std::max(1, omp_get_max_threads());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    if (std::FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long long period = 0;
        if (std::fscanf(f, "%31s %lld", quota, &period) == 2 && quota[0] != 'm' && period > 0) {
            const long long q = std::atoll(quota);
            if (q > 0) n = std::min<long long>(n, std::max<long long>(1, (q + period - 1) / period));
        }
        std::fclose(f);
    }
    return n;
}

HostEngine::HostEngine(const GridConfig& grid, int threads) : g_(grid) {
    threads_ = threads > 0 ? threads : usable_cpus();
    W_ = g_.width;
    H_ = g_.height;
    tiles_x_ = (W_ + g_.tile_width - 1) / g_.tile_width;
    tiles_y_ = (H_ + g_.tile_height - 1) / g_.tile_height;
    const int want = std::max(1, std::min(H_, threads_ * 4));       // a few stripes per thread: dynamic scheduling evens out clusters
    stripe_rows_ = (H_ + want - 1) / want;
    nstripes_ = (H_ + stripe_rows_ - 1) / stripe_rows_;
    touched_.assign((size_t)tiles_x_ * tiles_y_, 0u);
}

void HostEngine::init_planes(HostPlanes& p, uint32_t mask) const {
    p.mask = mask;
    const size_t cells = (size_t)W_ * H_;
    const float ident[4] = {0.0f, 0.0f, -FLT_MAX, FLT_MAX};            // builtin_ops.h: identity() of Sum / Count / Max / Min
    for (int k = 0; k < 4; ++k) {
        if (!(mask & (1u << k))) { std::vector<float>().swap(p.plane[k]); continue; }
        p.plane[k].resize(cells);
        float* dst = p.plane[k].data();
        const float v = ident[k];
#pragma omp parallel for num_threads(threads_) schedule(static)
        for (int64_t i = 0; i < (int64_t)cells; ++i) dst[i] = v;
    }
}

size_t HostEngine::route(const double* x, const double* y, const uint8_t* keep, size_t n) {
    x_ = x;
    y_ = y;
    n_ = n;
    col_.resize(n);
    row_.resize(n);
    // GridConfig::world_to_cell (src/core/grid_config.cpp:24-43): inclusive bounds (src/core/types.cpp:41-43), floor of a true
    // f64 division, clamp
    const double ox = g_.bounds.min_x, oy = g_.bounds.max_y, csx = g_.cell_size_x, csy = g_.cell_size_y;
    const double min_x = g_.bounds.min_x, max_x = g_.bounds.max_x, min_y = g_.bounds.min_y, max_y = g_.bounds.max_y;
    const int tw = g_.tile_width, th = g_.tile_height;
    size_t valid = 0;
    uint32_t* flags = touched_.data();
#pragma omp parallel for num_threads(threads_) schedule(static) reduction(+ : valid)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        int c = 0, r = -1;
        const double wx = x[i], wy = y[i];
        if ((!keep || keep[i]) && wx >= min_x && wx <= max_x && wy >= min_y && wy <= max_y) {
            c = static_cast<int>(std::floor((wx - ox) / csx));
            r = static_cast<int>(std::floor((wy - oy) / csy));
            c = std::max(0, std::min(c, W_ - 1));
            r = std::max(0, std::min(r, H_ - 1));
            // (every writer stores 1: a relaxed atomic store, so that threads marking the same tile do not race)
            uint32_t* f = flags + (size_t)(r / th) * tiles_x_ + (size_t)(c / tw);
            if (__atomic_load_n(f, __ATOMIC_RELAXED) == 0u) __atomic_store_n(f, 1u, __ATOMIC_RELAXED);
            ++valid;
        }
        col_[(size_t)i] = c;
        row_[(size_t)i] = r;
    }
    return valid;
}

// Point lists per stripe, ascending point index inside a stripe whatever the thread count: thread t lists the points of the
// t-th contiguous chunk of the cloud, and a stripe's list is the threads' parts in thread order.
template <class RowsOf>
void HostEngine::build_lists(std::vector<uint32_t>& list, std::vector<size_t>& first, RowsOf rows_of) const {
    const int T = threads_, S = nstripes_;
    const size_t chunk = (n_ + (size_t)T - 1) / (size_t)T;
    std::vector<size_t> cnt((size_t)T * S, 0);
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; ++t) {
        size_t* mine = cnt.data() + (size_t)t * S;
        const size_t i0 = std::min(n_, (size_t)t * chunk), i1 = std::min(n_, i0 + chunk);
        for (size_t i = i0; i < i1; ++i) {
            Span sp;
            if (!rows_of(i, sp)) continue;
            for (int s = sp.lo / stripe_rows_; s <= sp.hi / stripe_rows_; ++s) ++mine[s];
        }
    }
    first.assign((size_t)S + 1, 0);
    size_t run = 0;
    for (int s = 0; s < S; ++s) {
        first[(size_t)s] = run;
        for (int t = 0; t < T; ++t) {
            const size_t c = cnt[(size_t)t * S + s];
            cnt[(size_t)t * S + s] = run;
            run += c;
        }
    }
    first[(size_t)S] = run;
    list.resize(run);
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; ++t) {
        size_t* mine = cnt.data() + (size_t)t * S;
        const size_t i0 = std::min(n_, (size_t)t * chunk), i1 = std::min(n_, i0 + chunk);
        for (size_t i = i0; i < i1; ++i) {
            Span sp;
            if (!rows_of(i, sp)) continue;
            for (int s = sp.lo / stripe_rows_; s <= sp.hi / stripe_rows_; ++s) list[mine[s]++] = (uint32_t)i;
        }
    }
}

void HostEngine::scatter_point(HostPlanes& p, const float* v) {
    if (n_ == 0) return;
    std::vector<uint32_t> list;
    std::vector<size_t> first;
    build_lists(list, first, [&](size_t i, Span& sp) {
        if (row_[i] < 0) return false;
        sp.lo = sp.hi = row_[i];
        return true;
    });
    float* sum = (p.mask & 1u) ? p.plane[0].data() : nullptr;
    float* wgt = (p.mask & 2u) ? p.plane[1].data() : nullptr;
    float* mx = (p.mask & 4u) ? p.plane[2].data() : nullptr;
    float* mn = (p.mask & 8u) ? p.plane[3].data() : nullptr;
    // Op::combine(acc, v) of include/pcr/ops/builtin_ops.h:10-103, f32 like the reference's tile state
#pragma omp parallel for num_threads(threads_) schedule(dynamic, 1)
    for (int s = 0; s < nstripes_; ++s) {
        for (size_t k = first[(size_t)s]; k < first[(size_t)s + 1]; ++k) {
            const uint32_t i = list[k];
            const size_t cell = (size_t)row_[i] * W_ + (size_t)col_[i];
            const float val = v ? v[i] : 0.0f;
            if (sum) sum[cell] = sum[cell] + val;
            if (wgt) wgt[cell] = wgt[cell] + 1.0f;
            if (mx) mx[cell] = fmaxf(mx[cell], val);
            if (mn) mn[cell] = fminf(mn[cell], val);
        }
    }
}

void HostEngine::scatter_glyph(HostPlanes& p, const GlyphSpec& glyph, const HostGlyphArrays& arr, const float* v) {
    if (n_ == 0) return;
    float* sum = (p.mask & 1u) ? p.plane[0].data() : nullptr;
    float* wgt = (p.mask & 2u) ? p.plane[1].data() : nullptr;
    std::vector<uint32_t> list;
    std::vector<size_t> first;
    const GridConfig& g = g_;
    if (glyph.type == GlyphType::Gaussian) {
        build_lists(list, first, [&](size_t i, Span& sp) {
            if (row_[i] < 0) return false;
            const Splat s = make_splat(g, glyph, arr, i, x_[i], y_[i], col_[i], row_[i]);
            if (!s.ok) return false;
            sp.lo = (int)std::max<int64_t>((int64_t)s.icy - s.r, s.clip.r0);
            sp.hi = (int)std::min<int64_t>((int64_t)s.icy + s.r, s.clip.r1 - 1);
            return sp.lo <= sp.hi && (int64_t)s.icx + s.r >= s.clip.c0 && (int64_t)s.icx - s.r < s.clip.c1;
        });
#pragma omp parallel for num_threads(threads_) schedule(dynamic, 1)
        for (int st = 0; st < nstripes_; ++st) {
            const int srow0 = st * stripe_rows_, srow1 = std::min(H_, srow0 + stripe_rows_);
            for (size_t k = first[(size_t)st]; k < first[(size_t)st + 1]; ++k) {
                const uint32_t i = list[k];
                const Splat s = make_splat(g, glyph, arr, i, x_[i], y_[i], col_[i], row_[i]);
                const float val = v ? v[i] : 0.0f;
                // the footprint's cells inside the centre cell's reference tile AND this stripe; every cell's weight is the
                // reference's expression with identically rounded operands (glyph_kernels.cu:157-166)
                const int gy0 = (int)std::max<int64_t>((int64_t)s.icy - s.r, std::max(s.clip.r0, srow0));
                const int gy1 = (int)std::min<int64_t>((int64_t)s.icy + s.r, std::min(s.clip.r1, srow1) - 1);
                const int gx0 = (int)std::max<int64_t>((int64_t)s.icx - s.r, s.clip.c0);
                const int gx1 = (int)std::min<int64_t>((int64_t)s.icx + s.r, s.clip.c1 - 1);
                for (int gy = gy0; gy <= gy1; ++gy) {
                    const float rdy = static_cast<float>(gy - s.icy) - s.sub_cy;
                    for (int gx = gx0; gx <= gx1; ++gx) {
                        const float rdx = static_cast<float>(gx - s.icx) - s.sub_cx;
                        const float rx = rdx * s.cos_rot + rdy * (-s.sin_rot);
                        const float ry = rdx * s.sin_rot + rdy * s.cos_rot;
                        const float w = std::exp(-0.5f * ((rx / s.sx) * (rx / s.sx) + (ry / s.sy) * (ry / s.sy)));
                        if (w < 1e-6f) continue;
                        const size_t cell = (size_t)gy * W_ + (size_t)gx;
                        if (sum) sum[cell] += val * w;               // update_state_cpu, glyph_kernels.cu:36-74
                        if (wgt) wgt[cell] += w;
                    }
                }
            }
        }
        return;
    }
    if (glyph.type != GlyphType::Line) return;
    build_lists(list, first, [&](size_t i, Span& sp) {
        if (row_[i] < 0) return false;
        const Segment s = make_segment(g, glyph, arr, i, x_[i], y_[i], col_[i], row_[i]);
        if (!s.ok) return false;
        sp.lo = std::max(std::min(s.iy0, s.iy1), s.clip.r0);
        sp.hi = std::min(std::max(s.iy0, s.iy1), s.clip.r1 - 1);
        return sp.lo <= sp.hi;
    });
#pragma omp parallel for num_threads(threads_) schedule(dynamic, 1)
    for (int st = 0; st < nstripes_; ++st) {
        const int srow0 = st * stripe_rows_, srow1 = std::min(H_, srow0 + stripe_rows_);
        for (size_t k = first[(size_t)st]; k < first[(size_t)st + 1]; ++k) {
            const uint32_t i = list[k];
            const Segment s = make_segment(g, glyph, arr, i, x_[i], y_[i], col_[i], row_[i]);
            const float val = v ? v[i] : 0.0f;
            const int r0 = std::max(s.clip.r0, srow0), r1 = std::min(s.clip.r1, srow1);
            // the reference's walk, step for step (glyph_kernels.cu:252-278), weight 1 per visited cell
            const int ddx = std::abs(s.ix1 - s.ix0), ddy = std::abs(s.iy1 - s.iy0);
            const int stepx = s.ix0 < s.ix1 ? 1 : -1, stepy = s.iy0 < s.iy1 ? 1 : -1;
            int64_t err = (int64_t)ddx - ddy;
            int cx = s.ix0, cy = s.iy0;
            const int64_t max_steps = 2 * ((int64_t)ddx + ddy) + 2;
            for (int64_t step = 0; step <= max_steps; ++step) {
                if (cx >= s.clip.c0 && cx < s.clip.c1 && cy >= r0 && cy < r1) {
                    const size_t cell = (size_t)cy * W_ + (size_t)cx;
                    if (sum) sum[cell] += val * 1.0f;
                    if (wgt) wgt[cell] += 1.0f;
                }
                if (cx == s.ix1 && cy == s.iy1) break;
                const int64_t e2 = 2 * err;
                if (e2 > -ddy) { err -= ddy; cx += stepx; }
                if (e2 < ddx) { err += ddx; cy += stepy; }
            }
        }
    }
}

void HostEngine::finalize(const HostPlanes& p, ReductionType type, float* band) const {
    const float* sum = (p.mask & 1u) ? p.plane[0].data() : nullptr;
    const float* wgt = (p.mask & 2u) ? p.plane[1].data() : nullptr;
    const float* mx = (p.mask & 4u) ? p.plane[2].data() : nullptr;
    const float* mn = (p.mask & 8u) ? p.plane[3].data() : nullptr;
    const float nan = std::numeric_limits<float>::quiet_NaN();
    const int tw = g_.tile_width, th = g_.tile_height;
#pragma omp parallel for num_threads(threads_) schedule(static)
    for (int r = 0; r < H_; ++r) {
        const uint32_t* trow = touched_.data() + (size_t)(r / th) * tiles_x_;
        for (int c = 0; c < W_; ++c) {
            const size_t cell = (size_t)r * W_ + (size_t)c;
            float out = nan;                                         // a tile without state stays NaN (pipeline.cpp:1204-1222)
            if (trow[c / tw]) {
                switch (type) {                                      // Op::finalize, builtin_ops.h
                    case ReductionType::Sum: out = sum ? sum[cell] : nan; break;
                    case ReductionType::Count: out = wgt && wgt[cell] > 0.0f ? wgt[cell] : nan; break;
                    case ReductionType::Max: out = mx && mx[cell] != -FLT_MAX ? mx[cell] : nan; break;
                    case ReductionType::Min: out = mn && mn[cell] != FLT_MAX ? mn[cell] : nan; break;
                    case ReductionType::Average:
                    case ReductionType::WeightedAverage:
                        out = sum && wgt && wgt[cell] > 0.0f ? sum[cell] / wgt[cell] : nan;
                        break;
                    default: break;
                }
            }
            band[cell] = out;
        }
    }
}

size_t HostEngine::tiles_active() const {
    size_t k = 0;
    for (uint32_t t : touched_) k += t ? 1 : 0;
    return k;
}

}  // namespace detail
}  // namespace pcr
