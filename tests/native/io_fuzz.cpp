// ASan/UBSan harness for the file parsers (tests/test_io_sanitizers.py builds and runs it on the CPU): valid files in
// every layout, then random corruptions and truncations -- every call must return, never crash or over-allocate.
// argv[1] = scratch directory.
#include "pcr/core/grid.h"
#include "pcr/io/grid_io.h"
#include "pcr/io/point_cloud_io.h"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <vector>
using namespace pcr;
static std::vector<char> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return {std::istreambuf_iterator<char>(f), {}}; }
static void spit(const std::string& p, const std::vector<char>& b) { std::ofstream f(p, std::ios::binary | std::ios::trunc); f.write(b.data(), b.size()); }
int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    std::mt19937 rng(7);
    const int W = 150, H = 97;
    GridConfig cfg; cfg.bounds.min_x = 0; cfg.bounds.min_y = 0; cfg.bounds.max_x = W; cfg.bounds.max_y = H;
    cfg.cell_size_x = 1; cfg.cell_size_y = -1; cfg.width = W; cfg.height = H; cfg.tile_width = 64; cfg.tile_height = 48;
    cfg.crs = CRS::from_epsg(32618);
    std::vector<BandDesc> bands(2); bands[0].name = "a"; bands[1].name = "b<&>";
    auto g = Grid::create(W, H, bands);
    for (int b = 0; b < 2; ++b) for (int i = 0; i < W * H; ++i) g->band_f32(b)[i] = (i % 7 == 0) ? NAN : (float)(rng() % 1000) * 0.25f;
    long ok = 0, bad = 0;
    for (const char* comp : {"NONE", "LZW", "DEFLATE"}) for (int tiled = 0; tiled < 2; ++tiled) for (int big = 0; big < 2; ++big) {
        GeoTiffOptions o; o.compress = comp; o.bigtiff = big; o.tile_width = tiled ? 32 : 0; o.tile_height = tiled ? 48 : 0;
        const std::string p = dir + "/t.tif";
        Status s = write_geotiff(p, *g, cfg, o);
        if (!s.ok()) { std::printf("write failed %s\n", s.message.c_str()); return 1; }
        std::vector<float> out(W * H);
        for (int b = 0; b < 2; ++b) {
            s = read_geotiff_band(p, b, out.data(), W, H);
            if (!s.ok()) { std::printf("read failed %s\n", s.message.c_str()); return 1; }
            for (int i = 0; i < W * H; ++i) { float a = g->band_f32(b)[i]; if (!(a == out[i] || (a != a && out[i] != out[i]))) { std::printf("mismatch\n"); return 1; } }
        }
        std::vector<char> file = slurp(p);
        for (int trial = 0; trial < 120; ++trial) {
            std::vector<char> c = file;
            int nflip = 1 + rng() % 8;
            for (int k = 0; k < nflip; ++k) c[rng() % c.size()] = (char)rng();
            if (trial % 5 == 0) c.resize(rng() % c.size());
            spit(dir + "/c.tif", c);
            const std::string cf = dir + "/c.tif";
            int w, h, nb; CRS crs; BBox bb; std::vector<std::string> names;
            Status a = read_geotiff_info(cf, w, h, nb, crs, bb);
            Status b2 = read_geotiff_band(cf, 0, out.data(), W, H);
            (void)read_geotiff_band_names(cf, names);
            (a.ok() && b2.ok() ? ok : bad)++;
        }
    }
    // PCRP / CSV
    auto pc = PointCloud::create(500); pc->resize(500); pc->add_channel("v", DataType::Float32); pc->add_channel("k", DataType::Int32);
    for (int i = 0; i < 500; ++i) { pc->x()[i] = i * 0.5; pc->y()[i] = -i; pc->channel_f32("v")[i] = i; pc->channel_i32("k")[i] = i; }
    if (!write_point_cloud(dir + "/p.pcrp", *pc).ok() || !write_point_cloud(dir + "/p.csv", *pc, PointCloudFormat::CSV).ok()) return 1;
    for (int fmt = 0; fmt < 2; ++fmt) {
        const std::string path = dir + (fmt == 0 ? "/p.pcrp" : "/p.csv");
        std::vector<char> file = slurp(path);
        for (int trial = 0; trial < 200; ++trial) {
            std::vector<char> c = file;
            int nflip = 1 + rng() % 6;
            for (int k = 0; k < nflip; ++k) c[rng() % std::min<size_t>(c.size(), trial % 2 ? 64 : c.size())] = (char)rng();
            if (trial % 4 == 0) c.resize(rng() % c.size());
            const std::string q = dir + (fmt == 0 ? "/c.pcrp" : "/c.csv");
            spit(q, c);
            PointCloudInfo info;
            (void)read_point_cloud_info(q, info);
            auto r = read_point_cloud(q);
            auto rd = PointCloudReader::open(q);
            if (rd) { auto chunk = PointCloud::create(64); while (rd->read_chunk(*chunk, 64) > 0) {} }
            (r ? ok : bad)++;
        }
    }
    std::printf("parsers survived: %ld readable, %ld rejected\n", ok, bad);
    return 0;
}
