// grid_io.cpp -- GeoTIFF writer / reader without GDAL (see pcr/io/grid_io.h).
// Replaces src/io/grid_io.cpp:39-500 of the reference at the same API; argument checks and messages
// follow it where they exist ("grid must be on host", "grid dimensions mismatch config", ...).
// Formats: TIFF 6.0 (Adobe, 1992) sections 2, 8, 13 (LZW), 15 (tiles); BigTIFF (libtiff design);
// GeoTIFF 1.0 section 2.6-2.7 (tags 33550, 33922, 34264, 34735, 34737); GDAL's private tags 42112 / 42113.
#include "pcr/io/grid_io.h"

#include "pcr/core/grid.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <thread>

namespace pcr {

namespace {

enum : int { kCompNone = 1, kCompLzw = 5, kCompDeflate = 8 };
enum : uint16_t { tBYTE = 1, tASCII = 2, tSHORT = 3, tLONG = 4, tDOUBLE = 12, tLONG8 = 16 };

// ---- LZW, TIFF flavour: MSB-first codes of 9..12 bits, Clear = 256, EOI = 257, "early change" ----
class LzwEncoder {
public:
    void encode(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
        out_ = &out;
        acc_ = 0;
        nacc_ = 0;
        reset();
        put(256);
        if (n == 0) { put(257); flush(); return; }
        int w = src[0];
        for (size_t i = 1; i < n; ++i) {
            const int k = src[i];
            const uint32_t key = ((uint32_t)w << 8) | (uint32_t)k;
            uint32_t h = (key * 2654435761u) >> (32 - kHashBits);
            int found = -1;
            while (hkey_[h] != 0xFFFFFFFFu) {
                if (hkey_[h] == key) { found = hval_[h]; break; }
                h = (h + 1) & (kHashSize - 1);
            }
            if (found >= 0) { w = found; continue; }
            put(w);
            hkey_[h] = key;
            hval_[h] = (uint16_t)next_;
            grow();
            w = k;
        }
        put(w);
        grow();                 // the decoder adds an entry after every code, the last one included
        put(257);
        flush();
    }

private:
    static constexpr int kHashBits = 13, kHashSize = 1 << kHashBits;
    uint32_t hkey_[kHashSize];
    uint16_t hval_[kHashSize];
    int next_ = 258, nbits_ = 9;
    uint32_t acc_ = 0;
    int nacc_ = 0;
    std::vector<uint8_t>* out_ = nullptr;

    void reset() {
        std::fill(hkey_, hkey_ + kHashSize, 0xFFFFFFFFu);
        next_ = 258;
        nbits_ = 9;
    }
    // One table entry was added.  Codes widen after entries 511, 1023, 2047 (the decoder, one entry
    // behind, widens after 510, ... -- TIFF's "early change"); a full table is cleared.
    void grow() {
        ++next_;
        if (next_ == 4094) {
            put(256);
            reset();
        } else if (next_ == (1 << nbits_) && nbits_ < 12) {
            ++nbits_;
        }
    }
    void put(int code) {
        acc_ = (acc_ << nbits_) | (uint32_t)code;
        nacc_ += nbits_;
        while (nacc_ >= 8) {
            out_->push_back((uint8_t)(acc_ >> (nacc_ - 8)));
            nacc_ -= 8;
        }
    }
    void flush() {
        if (nacc_ > 0) out_->push_back((uint8_t)(acc_ << (8 - nacc_)));
        nacc_ = 0;
    }
};

bool lzw_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_n) {
    // table entry = (prefix code, last byte, length); strings are rebuilt backwards
    static thread_local uint16_t prefix[4096];
    static thread_local uint8_t suffix[4096];
    static thread_local uint16_t length[4096];
    static thread_local uint8_t first[4096];
    for (int i = 0; i < 256; ++i) { prefix[i] = 0xFFFF; suffix[i] = (uint8_t)i; length[i] = 1; first[i] = (uint8_t)i; }
    int next = 258, nbits = 9, old = -1;
    uint32_t acc = 0;
    int nacc = 0;
    size_t ip = 0, op = 0;
    auto emit = [&](int code) -> bool {
        const size_t len = length[code];
        if (op + len > dst_n) return false;
        size_t p = op + len;
        int c = code;
        while (c != 0xFFFF && p > op) { dst[--p] = suffix[c]; c = prefix[c]; }
        op += len;
        return true;
    };
    while (true) {
        while (nacc < nbits) {
            if (ip >= n) return op == dst_n;                   // tolerate a missing EOI
            acc = (acc << 8) | src[ip++];
            nacc += 8;
        }
        const int code = (int)((acc >> (nacc - nbits)) & ((1u << nbits) - 1));
        nacc -= nbits;
        if (code == 257) break;
        if (code == 256) { next = 258; nbits = 9; old = -1; continue; }
        if (old < 0) {
            if (code >= 256 || !emit(code)) return false;
            old = code;
            continue;
        }
        if (code < next) {
            if (!emit(code)) return false;
            if (next < 4096) { prefix[next] = (uint16_t)old; suffix[next] = first[code]; first[next] = first[old]; length[next] = (uint16_t)(length[old] + 1); ++next; }
        } else if (code == next && next < 4096) {
            prefix[next] = (uint16_t)old; suffix[next] = first[old]; first[next] = first[old]; length[next] = (uint16_t)(length[old] + 1); ++next;
            if (!emit(code)) return false;
        } else {
            return false;
        }
        if (next == (1 << nbits) - 1 && nbits < 12) ++nbits;
        old = code;
    }
    return op == dst_n;
}

// ---- TIFF output ------------------------------------------------------------------------------------
struct TiffSpec {
    int W = 0, H = 0, nb = 0;
    bool tiled = false;
    int bw = 0, bh = 0;               // block = tile, or strip of bh rows x W
    bool big = true;
    int compression = kCompLzw;
    int level = 6;
};

struct TagOut {
    uint16_t tag, type;
    uint64_t count;
    std::vector<uint8_t> data;
};

template <typename T>
void append(std::vector<uint8_t>& v, T x) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&x);
    v.insert(v.end(), p, p + sizeof(T));
}

std::string xml_escape(const std::string& s) {
    std::string o;
    for (char c : s) {
        switch (c) {
            case '&': o += "&amp;"; break;
            case '<': o += "&lt;"; break;
            case '>': o += "&gt;"; break;
            case '"': o += "&quot;"; break;
            default: o += c;
        }
    }
    return o;
}

class TiffOut {
public:
    ~TiffOut() { if (f_) std::fclose(f_); }

    Status open(const std::string& path, const TiffSpec& spec, const GridConfig& cfg, const std::vector<std::string>& names) {
        spec_ = spec;
        cfg_ = cfg;
        names_ = names;
        bxn_ = (spec.W + spec.bw - 1) / spec.bw;
        byn_ = (spec.H + spec.bh - 1) / spec.bh;
        offsets_.assign((size_t)bxn_ * byn_ * spec.nb, 0);
        counts_.assign(offsets_.size(), 0);
        f_ = std::fopen(path.c_str(), "wb");
        if (!f_) return Status::error(StatusCode::IoError, "failed to create GeoTIFF: " + path);
        uint8_t hdr[16] = {'I', 'I', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (spec.big) { hdr[2] = 43; hdr[4] = 8; }
        else hdr[2] = 42;
        pos_ = spec.big ? 16 : 8;
        if (std::fwrite(hdr, 1, pos_, f_) != pos_) return io_error();
        return Status::success();
    }

    int block_w() const { return spec_.bw; }
    int block_h() const { return spec_.bh; }

    // Block (bx, by) of `band` from src (top-left of the block's valid part, `stride` floats per row).
    Status write_block(int band, int bx, int by, const float* src, int64_t stride) {
        Status s = pack_block(bx, by, src, stride, raw_, packed_, lzw_);
        if (!s.ok()) return s;
        return append_block(((size_t)band * byn_ + by) * bxn_ + bx, packed_);
    }

    struct Job { int band, bx, by; const float* src; int64_t stride; };
    // Many blocks: compression runs on worker threads (blocks are independent), the file is appended in order.
    Status write_blocks(const std::vector<Job>& jobs) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t nthreads = spec_.compression == kCompNone ? 1 : std::min<size_t>({(size_t)hw, 32, jobs.size()});
        if (nthreads <= 1) {
            for (const Job& j : jobs) { Status s = write_block(j.band, j.bx, j.by, j.src, j.stride); if (!s.ok()) return s; }
            return Status::success();
        }
        const size_t batch = nthreads * 4;
        std::vector<std::vector<uint8_t>> packed(batch);
        std::vector<Status> st(batch);
        for (size_t j0 = 0; j0 < jobs.size(); j0 += batch) {
            const size_t nb = std::min(batch, jobs.size() - j0);
            std::atomic<size_t> next{0};
            auto work = [&]() {
                std::vector<float> raw;
                auto enc = std::make_unique<LzwEncoder>();
                for (size_t k = next++; k < nb; k = next++) {
                    const Job& j = jobs[j0 + k];
                    st[k] = pack_block(j.bx, j.by, j.src, j.stride, raw, packed[k], *enc);
                }
            };
            std::vector<std::thread> pool;
            for (size_t t = 0; t + 1 < std::min(nthreads, nb); ++t) pool.emplace_back(work);
            work();
            for (auto& t : pool) t.join();
            for (size_t k = 0; k < nb; ++k) {
                if (!st[k].ok()) return st[k];
                const Job& j = jobs[j0 + k];
                Status s = append_block(((size_t)j.band * byn_ + j.by) * bxn_ + j.bx, packed[k]);
                if (!s.ok()) return s;
            }
        }
        return Status::success();
    }

    Status close() {
        if (!f_) return Status::error(StatusCode::InvalidArgument, "writer not open");
        // blocks nobody wrote share one all-nodata block
        uint64_t fill_off[2] = {0, 0}, fill_cnt[2] = {0, 0};       // [0] full block, [1] short last strip
        for (size_t i = 0; i < offsets_.size(); ++i) {
            if (offsets_[i]) continue;
            const int by = (int)((i / bxn_) % byn_);
            const int rows = spec_.tiled ? spec_.bh : std::min(spec_.bh, spec_.H - by * spec_.bh);
            const int kind = rows == spec_.bh ? 0 : 1;
            if (!fill_off[kind]) {
                raw_.assign((size_t)spec_.bw * rows, std::nanf(""));
                Status s = compress_bytes(reinterpret_cast<const uint8_t*>(raw_.data()), raw_.size() * 4, packed_, lzw_);
                if (s.ok()) s = append_block(i, packed_);
                if (!s.ok()) return s;
                fill_off[kind] = offsets_[i];
                fill_cnt[kind] = counts_[i];
            } else {
                offsets_[i] = fill_off[kind];
                counts_[i] = fill_cnt[kind];
            }
        }
        Status s = write_directory();
        const bool bad = std::fclose(f_) != 0;
        f_ = nullptr;
        if (!s.ok()) return s;
        return bad ? Status::error(StatusCode::IoError, "failed to write GeoTIFF") : Status::success();
    }

private:
    TiffSpec spec_;
    GridConfig cfg_;
    std::vector<std::string> names_;
    FILE* f_ = nullptr;
    uint64_t pos_ = 0;
    int bxn_ = 0, byn_ = 0;
    std::vector<uint64_t> offsets_, counts_;
    std::vector<float> raw_;
    std::vector<uint8_t> packed_;
    LzwEncoder lzw_;

    Status io_error() { return Status::error(StatusCode::IoError, "failed to write GeoTIFF data"); }

    Status put_bytes(const void* p, size_t n) {
        if (n && std::fwrite(p, 1, n, f_) != n) return io_error();
        pos_ += n;
        if (pos_ & 1) { const uint8_t z = 0; if (std::fwrite(&z, 1, 1, f_) != 1) return io_error(); ++pos_; }
        return Status::success();
    }

    // thread-safe: touches only its arguments and the (constant) spec
    Status pack_block(int bx, int by, const float* src, int64_t stride, std::vector<float>& raw,
                      std::vector<uint8_t>& out, LzwEncoder& enc) const {
        const int cols = std::min(spec_.bw, spec_.W - bx * spec_.bw);
        const int rows = std::min(spec_.bh, spec_.H - by * spec_.bh);
        const int out_rows = spec_.tiled ? spec_.bh : rows;         // tiles are padded, the last strip is short
        raw.assign((size_t)spec_.bw * out_rows, std::nanf(""));
        for (int r = 0; r < rows; ++r) std::memcpy(&raw[(size_t)r * spec_.bw], src + (int64_t)r * stride, (size_t)cols * 4);
        return compress_bytes(reinterpret_cast<const uint8_t*>(raw.data()), raw.size() * 4, out, enc);
    }

    Status compress_bytes(const uint8_t* bytes, size_t n, std::vector<uint8_t>& out, LzwEncoder& enc) const {
        out.clear();
        if (spec_.compression == kCompLzw) {
            enc.encode(bytes, n, out);
        } else if (spec_.compression == kCompDeflate) {
            uLongf cap = compressBound((uLong)n);
            out.resize(cap);
            if (compress2(out.data(), &cap, bytes, (uLong)n, spec_.level) != Z_OK)
                return Status::error(StatusCode::IoError, "failed to compress GeoTIFF block");
            out.resize(cap);
        } else {
            out.assign(bytes, bytes + n);
        }
        return Status::success();
    }

    Status append_block(size_t index, const std::vector<uint8_t>& bytes) {
        if (!spec_.big && pos_ + bytes.size() > 0xFFFFFFF0ull)
            return Status::error(StatusCode::IoError, "GeoTIFF larger than 4 GB needs options.bigtiff");
        offsets_[index] = pos_;
        counts_[index] = bytes.size();
        return put_bytes(bytes.data(), bytes.size());
    }

    void add_shorts(std::vector<TagOut>& t, uint16_t tag, const std::vector<uint16_t>& v) {
        TagOut o{tag, tSHORT, v.size(), {}};
        for (uint16_t x : v) append(o.data, x);
        t.push_back(o);
    }
    void add_long(std::vector<TagOut>& t, uint16_t tag, uint32_t v) {
        TagOut o{tag, tLONG, 1, {}};
        append(o.data, v);
        t.push_back(o);
    }
    void add_doubles(std::vector<TagOut>& t, uint16_t tag, const std::vector<double>& v) {
        TagOut o{tag, tDOUBLE, v.size(), {}};
        for (double x : v) append(o.data, x);
        t.push_back(o);
    }
    void add_ascii(std::vector<TagOut>& t, uint16_t tag, const std::string& s) {
        TagOut o{tag, tASCII, s.size() + 1, {}};
        o.data.assign(s.begin(), s.end());
        o.data.push_back(0);
        t.push_back(o);
    }
    void add_offsets(std::vector<TagOut>& t, uint16_t tag, const std::vector<uint64_t>& v) {
        TagOut o{tag, (uint16_t)(spec_.big ? tLONG8 : tLONG), v.size(), {}};
        for (uint64_t x : v) { if (spec_.big) append(o.data, x); else append(o.data, (uint32_t)x); }
        t.push_back(o);
    }

    Status write_directory() {
        std::vector<TagOut> tags;
        const int nb = spec_.nb;
        add_long(tags, 256, (uint32_t)spec_.W);
        add_long(tags, 257, (uint32_t)spec_.H);
        add_shorts(tags, 258, std::vector<uint16_t>(nb, 32));
        add_shorts(tags, 259, {(uint16_t)spec_.compression});
        add_shorts(tags, 262, {1});                                    // BlackIsZero
        add_shorts(tags, 277, {(uint16_t)nb});
        add_shorts(tags, 284, {(uint16_t)(nb > 1 ? 2 : 1)});           // one plane per band
        if (spec_.tiled) {
            add_long(tags, 322, (uint32_t)spec_.bw);
            add_long(tags, 323, (uint32_t)spec_.bh);
            add_offsets(tags, 324, offsets_);
            add_offsets(tags, 325, counts_);
        } else {
            add_offsets(tags, 273, offsets_);
            add_long(tags, 278, (uint32_t)spec_.bh);
            add_offsets(tags, 279, counts_);
        }
        if (nb > 1) add_shorts(tags, 338, std::vector<uint16_t>(nb - 1, 0));   // extra samples: unspecified
        add_shorts(tags, 339, std::vector<uint16_t>(nb, 3));           // IEEE floating point

        // georeferencing: cell (col, row) corner -> world (gdal_geotransform of the reference)
        const double csx = cfg_.cell_size_x, csy = cfg_.cell_size_y;
        if (csy < 0.0 && csx > 0.0) {
            add_doubles(tags, 33550, {csx, -csy, 0.0});
            add_doubles(tags, 33922, {0.0, 0.0, 0.0, cfg_.bounds.min_x, cfg_.bounds.max_y, 0.0});
        } else {
            add_doubles(tags, 34264, {csx, 0, 0, cfg_.bounds.min_x, 0, csy, 0, cfg_.bounds.max_y, 0, 0, 0, 0, 0, 0, 0, 1});
        }
        std::vector<uint16_t> keys = {1, 1, 0, 0};
        std::string ascii;
        auto key = [&](uint16_t id, uint16_t loc, uint16_t cnt, uint16_t val) {
            keys.push_back(id); keys.push_back(loc); keys.push_back(cnt); keys.push_back(val);
            ++keys[3];
        };
        const bool geographic = cfg_.crs.is_valid() && cfg_.crs.is_geographic();
        key(1024, 0, 1, cfg_.crs.is_valid() ? (geographic ? 2 : 1) : 32767);   // GTModelType
        key(1025, 0, 1, 1);                                                  // RasterPixelIsArea
        if (cfg_.crs.epsg > 0 && cfg_.crs.epsg < 65536) {
            key(geographic ? 2048 : 3072, 0, 1, (uint16_t)cfg_.crs.epsg);
        } else if (!cfg_.crs.wkt.empty()) {
            std::string cite = cfg_.crs.wkt.substr(0, 4000);
            std::replace(cite.begin(), cite.end(), '|', ' ');
            key(1026, 34737, (uint16_t)(cite.size() + 1), 0);                 // GTCitation: the WKT itself
            ascii = cite + "|";
        }
        add_shorts(tags, 34735, keys);
        if (!ascii.empty()) add_ascii(tags, 34737, ascii);
        std::ostringstream md;
        md << "<GDALMetadata>\n";
        for (int b = 0; b < nb; ++b)
            if (b < (int)names_.size() && !names_[b].empty())
                md << "  <Item name=\"DESCRIPTION\" sample=\"" << b << "\" role=\"description\">" << xml_escape(names_[b]) << "</Item>\n";
        md << "</GDALMetadata>\n";
        add_ascii(tags, 42112, md.str());
        add_ascii(tags, 42113, "nan");

        std::sort(tags.begin(), tags.end(), [](const TagOut& a, const TagOut& b) { return a.tag < b.tag; });
        // out-of-line values first, then the directory
        const size_t inline_cap = spec_.big ? 8 : 4;
        std::vector<uint64_t> where(tags.size(), 0);
        for (size_t i = 0; i < tags.size(); ++i) {
            if (tags[i].data.size() <= inline_cap) continue;
            where[i] = pos_;
            Status s = put_bytes(tags[i].data.data(), tags[i].data.size());
            if (!s.ok()) return s;
        }
        const uint64_t ifd = pos_;
        std::vector<uint8_t> dir;
        if (spec_.big) append(dir, (uint64_t)tags.size());
        else append(dir, (uint16_t)tags.size());
        for (size_t i = 0; i < tags.size(); ++i) {
            append(dir, tags[i].tag);
            append(dir, tags[i].type);
            if (spec_.big) append(dir, (uint64_t)tags[i].count);
            else append(dir, (uint32_t)tags[i].count);
            std::vector<uint8_t> val(inline_cap, 0);
            if (tags[i].data.size() <= inline_cap) std::memcpy(val.data(), tags[i].data.data(), tags[i].data.size());
            else if (spec_.big) std::memcpy(val.data(), &where[i], 8);
            else { const uint32_t w = (uint32_t)where[i]; std::memcpy(val.data(), &w, 4); }
            dir.insert(dir.end(), val.begin(), val.end());
        }
        if (spec_.big) append(dir, (uint64_t)0);
        else append(dir, (uint32_t)0);
        if (!spec_.big && ifd + dir.size() > 0xFFFFFFF0ull)
            return Status::error(StatusCode::IoError, "GeoTIFF larger than 4 GB needs options.bigtiff");
        Status s = put_bytes(dir.data(), dir.size());
        if (!s.ok()) return s;
        if (std::fseek(f_, spec_.big ? 8 : 4, SEEK_SET) != 0) return io_error();
        if (spec_.big) { if (std::fwrite(&ifd, 8, 1, f_) != 1) return io_error(); }
        else { const uint32_t w = (uint32_t)ifd; if (std::fwrite(&w, 4, 1, f_) != 1) return io_error(); }
        return Status::success();
    }
};

Status make_spec(const GridConfig& cfg, int nb, const GeoTiffOptions& opt, TiffSpec* out) {
    if (opt.cloud_optimized)
        return Status::error(StatusCode::NotImplemented, "cloud-optimized GeoTIFF (overviews) is not implemented in this build");
    TiffSpec s;
    s.W = cfg.width;
    s.H = cfg.height;
    s.nb = nb;
    if (s.W <= 0 || s.H <= 0 || nb <= 0) return Status::error(StatusCode::InvalidArgument, "empty grid");
    std::string c = opt.compress;
    std::transform(c.begin(), c.end(), c.begin(), [](unsigned char ch) { return (char)std::toupper(ch); });
    if (c.empty() || c == "NONE") s.compression = kCompNone;
    else if (c == "LZW") s.compression = kCompLzw;
    else if (c == "DEFLATE") s.compression = kCompDeflate;
    else return Status::error(StatusCode::NotImplemented, "GeoTIFF compression not available in this build: " + opt.compress);
    s.level = std::max(1, std::min(9, opt.compress_level));
    s.big = opt.bigtiff;
    if (opt.tile_width > 0 && opt.tile_height > 0) {
        if (opt.tile_width % 16 || opt.tile_height % 16)
            return Status::error(StatusCode::InvalidArgument, "GeoTIFF tile size must be a multiple of 16");
        s.tiled = true;
        s.bw = opt.tile_width;
        s.bh = opt.tile_height;
    } else {
        s.tiled = false;
        s.bw = s.W;
        s.bh = std::max(1, std::min(s.H, 65536 / std::max(1, s.W)));     // ~256 KB strips
    }
    *out = s;
    return Status::success();
}

// ---- TIFF input --------------------------------------------------------------------------------------
struct TagIn {
    uint16_t type = 0;
    uint64_t count = 0;
    std::vector<uint8_t> data;
};

size_t type_size(uint16_t t) {
    switch (t) {
        case 1: case 2: case 6: case 7: return 1;
        case 3: case 8: return 2;
        case 4: case 9: case 11: case 13: return 4;
        case 5: case 10: case 12: case 16: case 17: case 18: return 8;
        default: return 0;
    }
}

struct TiffIn {
    FILE* f = nullptr;
    bool big = false;
    uint64_t file_size = 0;
    std::map<uint16_t, TagIn> tags;
    ~TiffIn() { if (f) std::fclose(f); }

    Status open(const std::string& path) {
        f = std::fopen(path.c_str(), "rb");
        if (!f) return Status::error(StatusCode::IoError, "failed to open file: " + path);
        uint8_t h[16];
        if (std::fread(h, 1, 8, f) != 8 || h[0] != 'I' || h[1] != 'I')
            return Status::error(StatusCode::IoError, "not a little-endian TIFF file: " + path);
        uint64_t ifd = 0;
        if (h[2] == 42) { uint32_t o; std::memcpy(&o, h + 4, 4); ifd = o; }
        else if (h[2] == 43) {
            big = true;
            if (std::fread(h + 8, 1, 8, f) != 8) return Status::error(StatusCode::IoError, "truncated TIFF header");
            std::memcpy(&ifd, h + 8, 8);
        } else return Status::error(StatusCode::IoError, "not a TIFF file: " + path);
        std::fseek(f, 0, SEEK_END);
        file_size = (uint64_t)std::ftell(f);
        if (ifd >= file_size || std::fseek(f, (long)ifd, SEEK_SET) != 0)
            return Status::error(StatusCode::IoError, "bad TIFF directory offset");
        uint64_t n = 0;
        if (big) { if (std::fread(&n, 8, 1, f) != 1) return bad(); }
        else { uint16_t n16; if (std::fread(&n16, 2, 1, f) != 1) return bad(); n = n16; }
        if (n > 4096) return bad();
        const size_t esz = big ? 20 : 12, cap = big ? 8 : 4;
        std::vector<uint8_t> dir(n * esz);
        if (n && std::fread(dir.data(), 1, dir.size(), f) != dir.size()) return bad();
        for (uint64_t i = 0; i < n; ++i) {
            const uint8_t* e = &dir[i * esz];
            uint16_t tag, type;
            std::memcpy(&tag, e, 2);
            std::memcpy(&type, e + 2, 2);
            uint64_t count = 0;
            if (big) std::memcpy(&count, e + 4, 8);
            else { uint32_t c; std::memcpy(&c, e + 4, 4); count = c; }
            const size_t ts = type_size(type);
            if (!ts || count > file_size / ts) continue;              // a value array cannot be larger than the file
            TagIn t;
            t.type = type;
            t.count = count;
            t.data.resize(ts * count);
            const uint8_t* v = e + (big ? 12 : 8);
            if (t.data.size() <= cap) std::memcpy(t.data.data(), v, t.data.size());
            else {
                uint64_t off = 0;
                if (big) std::memcpy(&off, v, 8);
                else { uint32_t o; std::memcpy(&o, v, 4); off = o; }
                if (std::fseek(f, (long)off, SEEK_SET) != 0 || std::fread(t.data.data(), 1, t.data.size(), f) != t.data.size())
                    return bad();
            }
            tags[tag] = std::move(t);
        }
        return Status::success();
    }
    static Status bad() { return Status::error(StatusCode::IoError, "corrupt TIFF directory"); }

    bool has(uint16_t tag) const { return tags.count(tag) != 0; }
    uint64_t uint_at(uint16_t tag, size_t i, uint64_t dflt = 0) const {
        auto it = tags.find(tag);
        if (it == tags.end() || i >= it->second.count) return dflt;
        const uint8_t* p = it->second.data.data();
        switch (it->second.type) {
            case 1: return p[i];
            case 3: { uint16_t v; std::memcpy(&v, p + 2 * i, 2); return v; }
            case 4: case 13: { uint32_t v; std::memcpy(&v, p + 4 * i, 4); return v; }
            case 16: case 18: { uint64_t v; std::memcpy(&v, p + 8 * i, 8); return v; }
            default: return dflt;
        }
    }
    std::vector<double> doubles(uint16_t tag) const {
        std::vector<double> v;
        auto it = tags.find(tag);
        if (it == tags.end() || it->second.type != tDOUBLE) return v;
        v.resize(it->second.count);
        std::memcpy(v.data(), it->second.data.data(), v.size() * 8);
        return v;
    }
    std::string ascii(uint16_t tag) const {
        auto it = tags.find(tag);
        if (it == tags.end() || it->second.type != tASCII) return {};
        std::string s(it->second.data.begin(), it->second.data.end());
        while (!s.empty() && s.back() == 0) s.pop_back();
        return s;
    }
};

struct TiffGeom {
    int W = 0, H = 0, nb = 0, planar = 1, compression = 1;
    bool tiled = false;
    int bw = 0, bh = 0, bxn = 0, byn = 0;
};

Status read_geom(const TiffIn& t, TiffGeom* g) {
    g->W = (int)t.uint_at(256, 0);
    g->H = (int)t.uint_at(257, 0);
    g->nb = (int)t.uint_at(277, 0, 1);
    g->planar = (int)t.uint_at(284, 0, 1);
    g->compression = (int)t.uint_at(259, 0, 1);
    if (g->W <= 0 || g->H <= 0 || g->nb <= 0) return Status::error(StatusCode::IoError, "TIFF without image dimensions");
    if (t.uint_at(258, 0, 1) != 32 || t.uint_at(339, 0, 1) != 3)
        return Status::error(StatusCode::NotImplemented, "only 32-bit floating point TIFF samples are supported");
    if (g->compression != kCompNone && g->compression != kCompLzw && g->compression != kCompDeflate && g->compression != 32946)
        return Status::error(StatusCode::NotImplemented, "unsupported TIFF compression " + std::to_string(g->compression));
    if (t.uint_at(317, 0, 1) != 1) return Status::error(StatusCode::NotImplemented, "TIFF predictors are not supported");
    g->tiled = t.has(322);
    if (g->tiled) { g->bw = (int)t.uint_at(322, 0); g->bh = (int)t.uint_at(323, 0); }
    else { g->bw = g->W; g->bh = (int)std::min<uint64_t>(t.uint_at(278, 0, (uint64_t)g->H), (uint64_t)g->H); }
    if (g->bw <= 0 || g->bh <= 0) return Status::error(StatusCode::IoError, "TIFF with empty blocks");
    g->bxn = (g->W + g->bw - 1) / g->bw;
    g->byn = (g->H + g->bh - 1) / g->bh;
    return Status::success();
}

}  // namespace

// ---- writer API ----------------------------------------------------------------------------------------
Status write_geotiff(const std::string& path, const Grid& grid, const GridConfig& config, const GeoTiffOptions& options) {
    if (grid.location() != MemoryLocation::Host && grid.location() != MemoryLocation::HostPinned)
        return Status::error(StatusCode::InvalidArgument, "grid must be on host");
    if (grid.cols() != config.width || grid.rows() != config.height)
        return Status::error(StatusCode::InvalidArgument, "grid dimensions mismatch config");
    TiffSpec spec;
    Status s = make_spec(config, grid.num_bands(), options, &spec);
    if (!s.ok()) return s;
    std::vector<std::string> names;
    for (int b = 0; b < grid.num_bands(); ++b) names.push_back(grid.band_desc(b).name);
    TiffOut out;
    if (!(s = out.open(path, spec, config, names)).ok()) return s;
    const int bxn = (spec.W + spec.bw - 1) / spec.bw, byn = (spec.H + spec.bh - 1) / spec.bh;
    std::vector<TiffOut::Job> jobs;
    for (int b = 0; b < grid.num_bands(); ++b) {
        const float* band = grid.band_f32(b);
        if (!band) return Status::error(StatusCode::IoError, "failed to get band " + std::to_string(b));
        for (int by = 0; by < byn; ++by)
            for (int bx = 0; bx < bxn; ++bx)
                jobs.push_back({b, bx, by, band + (int64_t)by * spec.bh * spec.W + (int64_t)bx * spec.bw, spec.W});
    }
    if (!(s = out.write_blocks(jobs)).ok()) return s;
    return out.close();
}

struct TiledGeoTiffWriter::Impl {
    TiffOut out;
    TiffSpec spec;
    GridConfig config;
    int num_bands = 0;
    bool open = false;
    bool aligned = false;                 // reference tiles are unions of whole TIFF blocks: stream them
    std::vector<float> whole;             // otherwise: the image is assembled in memory and written at close
};

TiledGeoTiffWriter::~TiledGeoTiffWriter() {
    if (impl_ && impl_->open) (void)close();
}

std::unique_ptr<TiledGeoTiffWriter> TiledGeoTiffWriter::open(const std::string& path, const GridConfig& config,
                                                             const std::vector<std::string>& band_names,
                                                             const GeoTiffOptions& options) {
    auto w = std::unique_ptr<TiledGeoTiffWriter>(new TiledGeoTiffWriter());
    w->impl_ = std::make_unique<Impl>();
    Impl& s = *w->impl_;
    s.config = config;
    s.num_bands = (int)band_names.size();
    if (!make_spec(config, s.num_bands, options, &s.spec).ok()) return nullptr;
    const int tiles_x = (config.width + config.tile_width - 1) / config.tile_width;
    const int tiles_y = (config.height + config.tile_height - 1) / config.tile_height;
    s.aligned = (tiles_x == 1 || config.tile_width % s.spec.bw == 0) && (tiles_y == 1 || config.tile_height % s.spec.bh == 0) &&
                (s.spec.tiled || tiles_x == 1);
    if (!s.aligned) s.whole.assign((size_t)config.width * config.height * s.num_bands, std::nanf(""));
    if (!s.out.open(path, s.spec, config, band_names).ok()) return nullptr;
    s.open = true;
    return w;
}

Status TiledGeoTiffWriter::write_tile(TileIndex tile, const float* data, int num_bands) {
    if (!impl_ || !impl_->open) return Status::error(StatusCode::InvalidArgument, "writer not open");
    Impl& s = *impl_;
    if (num_bands != s.num_bands) return Status::error(StatusCode::InvalidArgument, "band count mismatch");
    if (!data) return Status::error(StatusCode::InvalidArgument, "null data pointer");
    int c0, r0, cols, rows;
    s.config.tile_cell_range(tile, c0, r0, cols, rows);
    if (cols <= 0 || rows <= 0) return Status::error(StatusCode::InvalidArgument, "tile outside the grid");
    const int64_t tile_cells = (int64_t)cols * rows;
    for (int b = 0; b < num_bands; ++b) {
        const float* src = data + b * tile_cells;
        if (!s.aligned) {
            float* dst = s.whole.data() + (size_t)b * s.config.width * s.config.height;
            for (int r = 0; r < rows; ++r)
                std::memcpy(dst + (size_t)(r0 + r) * s.config.width + c0, src + (size_t)r * cols, (size_t)cols * 4);
            continue;
        }
        for (int by = r0 / s.spec.bh; by * s.spec.bh < r0 + rows; ++by)
            for (int bx = c0 / s.spec.bw; bx * s.spec.bw < c0 + cols; ++bx) {
                Status st = s.out.write_block(b, bx, by, src + (int64_t)(by * s.spec.bh - r0) * cols + (bx * s.spec.bw - c0), cols);
                if (!st.ok()) return st;
            }
    }
    return Status::success();
}

Status TiledGeoTiffWriter::close() {
    if (!impl_ || !impl_->open) return Status::error(StatusCode::InvalidArgument, "writer not open");
    Impl& s = *impl_;
    s.open = false;
    if (!s.aligned) {
        const int bxn = (s.spec.W + s.spec.bw - 1) / s.spec.bw, byn = (s.spec.H + s.spec.bh - 1) / s.spec.bh;
        for (int b = 0; b < s.num_bands; ++b) {
            const float* band = s.whole.data() + (size_t)b * s.spec.W * s.spec.H;
            for (int by = 0; by < byn; ++by)
                for (int bx = 0; bx < bxn; ++bx) {
                    Status st = s.out.write_block(b, bx, by, band + (int64_t)by * s.spec.bh * s.spec.W + (int64_t)bx * s.spec.bw, s.spec.W);
                    if (!st.ok()) return st;
                }
        }
        s.whole.clear();
        s.whole.shrink_to_fit();
    }
    return s.out.close();
}

// ---- reader API ----------------------------------------------------------------------------------------
Status read_geotiff_info(const std::string& path, int& width, int& height, int& num_bands, CRS& crs, BBox& bounds) {
    TiffIn t;
    Status s = t.open(path);
    if (!s.ok()) return s;
    TiffGeom g;
    if (!(s = read_geom(t, &g)).ok()) return s;
    width = g.W;
    height = g.H;
    num_bands = g.nb;
    double gt[6] = {0, 1, 0, 0, 0, 1};
    bool have_gt = false;
    const auto scale = t.doubles(33550), tie = t.doubles(33922), xf = t.doubles(34264);
    if (scale.size() >= 2 && tie.size() >= 6) {
        gt[1] = scale[0];
        gt[5] = -scale[1];
        gt[0] = tie[3] - tie[0] * gt[1];
        gt[3] = tie[4] - tie[1] * gt[5];
        have_gt = true;
    } else if (xf.size() >= 16) {
        gt[0] = xf[3]; gt[1] = xf[0]; gt[2] = xf[1]; gt[3] = xf[7]; gt[4] = xf[4]; gt[5] = xf[5];
        have_gt = true;
    }
    if (have_gt) {                                                      // as upstream, grid_io.cpp:408-424
        bounds.min_x = gt[0];
        bounds.max_y = gt[3];
        bounds.max_x = gt[0] + gt[1] * width;
        bounds.min_y = gt[3] + gt[5] * height;
    }
    crs = CRS();
    auto kit = t.tags.find(34735);
    if (kit != t.tags.end() && kit->second.type == tSHORT && kit->second.count >= 4) {
        const size_t nkeys = (size_t)t.uint_at(34735, 3);
        const std::string params = t.ascii(34737);
        for (size_t k = 0; k < nkeys && 4 * (k + 1) + 3 < kit->second.count; ++k) {
            const uint64_t id = t.uint_at(34735, 4 * (k + 1)), loc = t.uint_at(34735, 4 * (k + 1) + 1);
            const uint64_t cnt = t.uint_at(34735, 4 * (k + 1) + 2), val = t.uint_at(34735, 4 * (k + 1) + 3);
            if ((id == 3072 || id == 2048) && loc == 0 && val > 0 && val < 32767) crs = CRS::from_epsg((int)val);
            if (id == 1026 && loc == 34737 && crs.epsg == 0 && val + cnt <= params.size() + 1 && cnt > 0) {
                std::string w = params.substr(val, cnt - 1);
                if (w.rfind("PROJ", 0) == 0 || w.rfind("GEOG", 0) == 0 || w.rfind("BOUNDCRS", 0) == 0) crs.wkt = w;
            }
        }
    }
    return Status::success();
}

Status read_geotiff_band(const std::string& path, int band_index, float* data, int width, int height) {
    if (!data) return Status::error(StatusCode::InvalidArgument, "null data pointer");
    if (band_index < 0) return Status::error(StatusCode::InvalidArgument, "invalid band index");
    TiffIn t;
    Status s = t.open(path);
    if (!s.ok()) return s;
    TiffGeom g;
    if (!(s = read_geom(t, &g)).ok()) return s;
    if (g.W != width || g.H != height) return Status::error(StatusCode::InvalidArgument, "dimension mismatch");
    if (band_index >= g.nb) return Status::error(StatusCode::InvalidArgument, "band index out of range");
    const uint16_t off_tag = g.tiled ? 324 : 273, cnt_tag = g.tiled ? 325 : 279;
    const int spp = g.planar == 2 ? 1 : g.nb;                          // samples per pixel inside a block
    std::vector<uint8_t> packed, raw;
    for (int by = 0; by < g.byn; ++by)
        for (int bx = 0; bx < g.bxn; ++bx) {
            const size_t idx = (g.planar == 2 ? (size_t)band_index * g.byn * g.bxn : 0) + (size_t)by * g.bxn + bx;
            const uint64_t off = t.uint_at(off_tag, idx), cnt = t.uint_at(cnt_tag, idx);
            const int rows = std::min(g.bh, g.H - by * g.bh), cols = std::min(g.bw, g.W - bx * g.bw);
            const int blk_rows = g.tiled ? g.bh : rows;
            const size_t raw_n = (size_t)g.bw * blk_rows * spp * 4;
            if (!off || !cnt || off > t.file_size || cnt > t.file_size - off)
                return Status::error(StatusCode::IoError, "failed to read band data");
            packed.resize(cnt);
            if (std::fseek(t.f, (long)off, SEEK_SET) != 0 || std::fread(packed.data(), 1, cnt, t.f) != cnt)
                return Status::error(StatusCode::IoError, "failed to read band data");
            const uint8_t* px = packed.data();
            if (g.compression == kCompNone) {
                if (cnt < raw_n) return Status::error(StatusCode::IoError, "failed to read band data");
            } else {
                raw.resize(raw_n);
                bool ok;
                if (g.compression == kCompLzw) ok = lzw_decode(packed.data(), cnt, raw.data(), raw_n);
                else { uLongf n = (uLongf)raw_n; ok = uncompress(raw.data(), &n, packed.data(), (uLong)cnt) == Z_OK && n == raw_n; }
                if (!ok) return Status::error(StatusCode::IoError, "failed to decompress band data");
                px = raw.data();
            }
            for (int r = 0; r < rows; ++r) {
                float* dst = data + (size_t)(by * g.bh + r) * g.W + (size_t)bx * g.bw;
                const uint8_t* src = px + ((size_t)r * g.bw * spp) * 4;
                if (spp == 1) std::memcpy(dst, src, (size_t)cols * 4);
                else for (int c = 0; c < cols; ++c) std::memcpy(dst + c, src + ((size_t)c * spp + band_index) * 4, 4);
            }
        }
    return Status::success();
}

Status read_geotiff_band_names(const std::string& path, std::vector<std::string>& names) {
    TiffIn t;
    Status s = t.open(path);
    if (!s.ok()) return s;
    TiffGeom g;
    if (!(s = read_geom(t, &g)).ok()) return s;
    names.assign(g.nb, "");
    const std::string md = t.ascii(42112);
    size_t p = 0;
    while ((p = md.find("<Item name=\"DESCRIPTION\"", p)) != std::string::npos) {
        const size_t sa = md.find("sample=\"", p), gt = md.find('>', p), end = md.find("</Item>", p);
        if (sa == std::string::npos || gt == std::string::npos || end == std::string::npos || sa > gt) break;
        const int b = std::atoi(md.c_str() + sa + 8);
        std::string v = md.substr(gt + 1, end - gt - 1);
        for (const auto& rep : {std::pair<const char*, const char*>{"&lt;", "<"}, {"&gt;", ">"}, {"&quot;", "\""}, {"&amp;", "&"}}) {
            size_t q = 0;
            while ((q = v.find(rep.first, q)) != std::string::npos) { v.replace(q, std::strlen(rep.first), rep.second); q += 1; }
        }
        if (b >= 0 && b < g.nb) names[b] = v;
        p = end;
    }
    return Status::success();
}

}  // namespace pcr
