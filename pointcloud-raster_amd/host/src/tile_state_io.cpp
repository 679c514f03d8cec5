// tile_state_io.cpp -- `.pcrt` tile-state files.  Behavioural contract and byte layout: the reference's
// src/io/tile_state_io.cpp:14-211 (same error classes and messages).
#include "pcr/io/tile_state_io.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>

namespace pcr {

namespace {

constexpr uint32_t kMagic = 0x54524350u;      // 'P','C','R','T' little-endian
constexpr uint32_t kVersion = 1;
constexpr size_t kHeaderBytes = 36;

struct FileCloser {
    void operator()(std::FILE* f) const { if (f) std::fclose(f); }
};
using File = std::unique_ptr<std::FILE, FileCloser>;

void put_u32(unsigned char* p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = v >> 24; }
uint32_t get_u32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

}  // namespace

Status write_tile_state(const std::string& path, TileIndex tile, int cols, int rows, int state_floats,
                        ReductionType type, const float* state) {
    if (!state) return Status::error(StatusCode::InvalidArgument, "null state pointer");
    if (cols <= 0 || rows <= 0 || state_floats <= 0) return Status::error(StatusCode::InvalidArgument, "invalid dimensions");
    unsigned char h[kHeaderBytes];
    std::memset(h, 0, sizeof h);
    put_u32(h + 0, kMagic);
    put_u32(h + 4, kVersion);
    put_u32(h + 8, static_cast<uint32_t>(tile.row));
    put_u32(h + 12, static_cast<uint32_t>(tile.col));
    put_u32(h + 16, static_cast<uint32_t>(cols));
    put_u32(h + 20, static_cast<uint32_t>(rows));
    put_u32(h + 24, static_cast<uint32_t>(state_floats));
    h[28] = static_cast<unsigned char>(type);
    File f(std::fopen(path.c_str(), "wb"));
    if (!f) return Status::error(StatusCode::IoError, "failed to open file for writing: " + path);
    if (std::fwrite(h, 1, sizeof h, f.get()) != sizeof h) return Status::error(StatusCode::IoError, "failed to write header");
    const size_t n = static_cast<size_t>(state_floats) * cols * rows;
    if (std::fwrite(state, sizeof(float), n, f.get()) != n) return Status::error(StatusCode::IoError, "failed to write state data");
    return Status::success();
}

Status read_tile_state_header(const std::string& path, TileIndex& tile, int& cols, int& rows,
                              int& state_floats, ReductionType& type) {
    File f(std::fopen(path.c_str(), "rb"));
    if (!f) return Status::error(StatusCode::IoError, "file not found: " + path);
    unsigned char h[kHeaderBytes];
    if (std::fread(h, 1, sizeof h, f.get()) != sizeof h) return Status::error(StatusCode::IoError, "failed to read header");
    if (get_u32(h) != kMagic) return Status::error(StatusCode::IoError, "invalid magic number (not a PCRT file)");
    const uint32_t ver = get_u32(h + 4);
    if (ver != kVersion)
        return Status::error(StatusCode::IoError, "unsupported version " + std::to_string(ver) + " (expected 1)");
    const int c = static_cast<int32_t>(get_u32(h + 16)), r = static_cast<int32_t>(get_u32(h + 20));
    const int k = static_cast<int32_t>(get_u32(h + 24));
    if (c <= 0 || r <= 0 || k <= 0) return Status::error(StatusCode::IoError, "invalid dimensions in header");
    tile.row = static_cast<int32_t>(get_u32(h + 8));
    tile.col = static_cast<int32_t>(get_u32(h + 12));
    cols = c;
    rows = r;
    state_floats = k;
    type = static_cast<ReductionType>(h[28]);
    return Status::success();
}

Status read_tile_state(const std::string& path, TileIndex& tile, int& cols, int& rows, int& state_floats,
                       ReductionType& type, float* state) {
    if (!state) return Status::error(StatusCode::InvalidArgument, "null state pointer");
    Status s = read_tile_state_header(path, tile, cols, rows, state_floats, type);
    if (!s.ok()) return s;
    File f(std::fopen(path.c_str(), "rb"));
    if (!f) return Status::error(StatusCode::IoError, "failed to open file: " + path);
    if (std::fseek(f.get(), static_cast<long>(kHeaderBytes), SEEK_SET) != 0)
        return Status::error(StatusCode::IoError, "failed to seek past header");
    const size_t n = static_cast<size_t>(state_floats) * cols * rows;
    if (std::fread(state, sizeof(float), n, f.get()) != n)
        return Status::error(StatusCode::IoError, "incomplete state data (file truncated?)");
    return Status::success();
}

std::string tile_state_filename(const std::string& dir, TileIndex tile) {
    char name[64];
    std::snprintf(name, sizeof name, "tile_%04d_%04d.pcrt", tile.row, tile.col);
    std::string out = dir;
    if (!out.empty() && out.back() != '/') out += '/';
    return out + name;
}

}  // namespace pcr
