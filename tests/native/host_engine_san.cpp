// ASan/UBSan harness for the host engine (tests/test_host_engine_sanitizers.py builds and runs it on the CPU): clouds with
// non-finite and enormous coordinates, values and glyph channels, footprints that leave the grid on every side, tiles that do
// not divide the grid, thread counts that exceed the rows -- every call must return without an out-of-bounds access, a signed
// overflow or a float-to-int conversion out of range, and the same input must give the same bits at 1 and at 7 threads.
#include "host_engine.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <vector>
using namespace pcr;
using namespace pcr::detail;

static GridConfig grid(int W, int H, int tw, int th, double cs, double x0, double y0) {
    GridConfig g;
    g.bounds.min_x = x0; g.bounds.min_y = y0; g.bounds.max_x = x0 + W * cs; g.bounds.max_y = y0 + H * cs;
    g.cell_size_x = cs; g.cell_size_y = -cs; g.width = W; g.height = H; g.tile_width = tw; g.tile_height = th;
    g.tiles_x = (W + tw - 1) / tw; g.tiles_y = (H + th - 1) / th;
    return g;
}

int main() {
    std::mt19937_64 rng(11);
    const float nanf_ = std::numeric_limits<float>::quiet_NaN(), inff = std::numeric_limits<float>::infinity();
    const double nand = std::numeric_limits<double>::quiet_NaN(), infd = std::numeric_limits<double>::infinity();
    long runs = 0;
    for (int trial = 0; trial < 21; ++trial) {
        const int W = 1 + (int)(rng() % 200), H = 1 + (int)(rng() % 150);
        const int tw = 1 + (int)(rng() % 90), th = 1 + (int)(rng() % 90);
        const double cs = (trial % 3 == 0) ? 0.37 : (trial % 3 == 1) ? 1.0 : 25.0;
        const GridConfig g = grid(W, H, tw, th, cs, trial % 2 ? -1.0e6 : 3.25, trial % 5 ? 7.5 : 1.0e7);
        const size_t n = 200 + rng() % 3000;
        std::vector<double> x(n), y(n);
        std::vector<float> v(n), dir(n), hl(n), sx(n), sy(n), rot(n);
        std::vector<uint8_t> keep(n);
        std::uniform_real_distribution<double> ux(g.bounds.min_x - 3 * cs, g.bounds.max_x + 3 * cs), uy(g.bounds.min_y - 3 * cs, g.bounds.max_y + 3 * cs);
        std::uniform_real_distribution<float> uf(-10.f, 10.f);
        const float odd_f[] = {nanf_, inff, -inff, 3.4e38f, -3.4e38f, 1e-45f, 0.f, -0.f, 1e30f, -1e30f};
        const double odd_d[] = {nand, infd, -infd, 1e308, -1e308, g.bounds.min_x, g.bounds.max_x, g.bounds.min_y, g.bounds.max_y, 0.0};
        for (size_t i = 0; i < n; ++i) {
            x[i] = ux(rng); y[i] = uy(rng);
            v[i] = uf(rng); dir[i] = uf(rng); hl[i] = uf(rng) * (float)cs; sx[i] = uf(rng) * (float)cs; sy[i] = uf(rng) * (float)cs; rot[i] = uf(rng);
            keep[i] = rng() % 5 != 0;
            if (rng() % 12 == 0) x[i] = odd_d[rng() % 10];
            if (rng() % 12 == 0) y[i] = odd_d[rng() % 10];
            if (rng() % 10 == 0) v[i] = odd_f[rng() % 10];
            if (rng() % 10 == 0) dir[i] = odd_f[rng() % 10];
            if (rng() % 10 == 0) hl[i] = odd_f[rng() % 10];
            if (rng() % 10 == 0) sx[i] = odd_f[rng() % 10];
            if (rng() % 10 == 0) sy[i] = odd_f[rng() % 10];
            if (rng() % 10 == 0) rot[i] = odd_f[rng() % 10];
        }
        GlyphSpec gauss, line;
        gauss.type = GlyphType::Gaussian;
        gauss.default_sigma_x = (float)cs * (trial % 4 == 0 ? 30.f : 1.5f);
        gauss.default_sigma_y = (float)cs * 0.8f;
        gauss.default_rotation = 0.4f;
        gauss.max_radius_cells = trial % 7 == 0 ? 1e9f : trial % 7 == 1 ? nanf_ : 6.f;
        line.type = GlyphType::Line;
        line.default_half_length = (float)cs * (trial % 6 == 0 ? 1e12f : 9.f);
        line.default_direction = 1.1f;
        line.max_radius_cells = trial % 5 == 0 ? inff : 12.f;
        std::vector<float> first;
        for (int threads : {1, 7, 400}) {
            HostEngine e(g, threads);
            HostPlanes pt, ga, gb, la, lb;
            e.init_planes(pt, 15u);
            e.init_planes(ga, 3u); e.init_planes(gb, 3u); e.init_planes(la, 3u); e.init_planes(lb, 2u);
            for (int round = 0; round < 2; ++round) {
                e.route(x.data(), y.data(), round ? keep.data() : nullptr, n);
                e.scatter_point(pt, v.data());
                HostGlyphArrays none, all;
                all.direction = dir.data(); all.half_length = hl.data(); all.sigma_x = sx.data(); all.sigma_y = sy.data(); all.rotation = rot.data();
                e.scatter_glyph(ga, gauss, none, v.data());
                e.scatter_glyph(gb, gauss, all, v.data());
                e.scatter_glyph(la, line, none, v.data());
                e.scatter_glyph(lb, line, all, nullptr);
            }
            std::vector<float> band((size_t)W * H), all_bands;
            const HostPlanes* planes[5] = {&pt, &ga, &gb, &la, &lb};
            const ReductionType types[6] = {ReductionType::Sum, ReductionType::Count, ReductionType::Average, ReductionType::WeightedAverage,
                                            ReductionType::Max, ReductionType::Min};
            for (int p = 0; p < 5; ++p)
                for (int t = 0; t < (p == 0 ? 6 : p == 4 ? 2 : 4); ++t) {
                    e.finalize(*planes[p], p == 4 ? (t ? ReductionType::Count : ReductionType::Sum) : types[t], band.data());
                    all_bands.insert(all_bands.end(), band.begin(), band.end());
                }
            if (first.empty()) first = all_bands;
            else if (std::memcmp(first.data(), all_bands.data(), first.size() * sizeof(float)) != 0) {
                std::printf("trial %d: %d threads give different bits\n", trial, threads);
                return 1;
            }
            ++runs;
        }
    }
    std::printf("host engine survived %ld adversarial runs\n", runs);
    return 0;
}
