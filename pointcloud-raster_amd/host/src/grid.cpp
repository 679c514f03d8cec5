// grid.cpp -- multi-band raster container.  Contract: the reference's src/core/grid.cpp
// (one allocation per band, row-major, Float32 typed accessors return nullptr on a dtype or
// index mismatch); Device grids are implemented here (the reference returns NotImplemented).
#include "pcr/core/grid.h"

#include "buffer.h"

#include <cmath>

namespace pcr {

struct Grid::Impl {
    int cols = 0, rows = 0;
    MemoryLocation loc = MemoryLocation::Host;         // what location() reports
    std::vector<BandDesc> descs;
    std::vector<detail::Buffer> bands;
};

Grid::~Grid() = default;

std::unique_ptr<Grid> Grid::create(int cols, int rows, const std::vector<BandDesc>& bands, MemoryLocation loc) {
    if (cols <= 0 || rows <= 0 || bands.empty()) return nullptr;
    auto g = std::unique_ptr<Grid>(new Grid());
    g->impl_ = std::make_unique<Impl>();
    Impl& m = *g->impl_;
    m.cols = cols;
    m.rows = rows;
    m.loc = loc;
    m.descs = bands;
    m.bands.resize(bands.size());
    for (size_t i = 0; i < bands.size(); ++i) {
        size_t bytes = static_cast<size_t>(cols) * rows * data_type_size(bands[i].dtype);
        if (!m.bands[i].allocate(bytes, loc).ok()) return nullptr;
    }
    return g;
}

std::unique_ptr<Grid> Grid::create_host_page_locked(int cols, int rows, const std::vector<BandDesc>& bands) {
    auto g = create(cols, rows, bands, MemoryLocation::HostPinned);
    if (g) g->impl_->loc = MemoryLocation::Host;
    return g;
}

std::unique_ptr<Grid> Grid::create_for_tile(const GridConfig& config, TileIndex tile,
                                            const std::vector<BandDesc>& bands, MemoryLocation loc) {
    int c0, r0, nc, nr;
    config.tile_cell_range(tile, c0, r0, nc, nr);
    return create(nc, nr, bands, loc);
}

int Grid::num_bands() const { return impl_ ? static_cast<int>(impl_->descs.size()) : 0; }

BandDesc Grid::band_desc(int i) const {
    if (!impl_ || i < 0 || i >= num_bands()) return BandDesc{};
    return impl_->descs[i];
}

int Grid::band_index(const std::string& name) const {
    for (int i = 0; i < num_bands(); ++i)
        if (impl_->descs[i].name == name) return i;
    return -1;
}

void* Grid::band_data(int i) {
    if (!impl_ || i < 0 || i >= num_bands()) return nullptr;
    return impl_->bands[i].data();
}
const void* Grid::band_data(int i) const { return const_cast<Grid*>(this)->band_data(i); }

float* Grid::band_f32(int i) {
    if (!impl_ || i < 0 || i >= num_bands() || impl_->descs[i].dtype != DataType::Float32) return nullptr;
    return static_cast<float*>(impl_->bands[i].data());
}
const float* Grid::band_f32(int i) const { return const_cast<Grid*>(this)->band_f32(i); }
float* Grid::band_f32(const std::string& name) { return band_f32(band_index(name)); }
const float* Grid::band_f32(const std::string& name) const { return band_f32(band_index(name)); }

int Grid::cols() const { return impl_ ? impl_->cols : 0; }
int Grid::rows() const { return impl_ ? impl_->rows : 0; }
int64_t Grid::cell_count() const { return static_cast<int64_t>(cols()) * rows(); }
MemoryLocation Grid::location() const { return impl_ ? impl_->loc : MemoryLocation::Host; }

Status Grid::fill_band(int i, float value) {
    float* p = band_f32(i);
    if (!p) return Status::error(StatusCode::InvalidArgument, "Invalid band index or data type");
    if (impl_->bands[i].location() == MemoryLocation::Device) {
        detail::hip_status(pcr_hip_plane_fill(p, value, cell_count(), nullptr));
        return detail::hip_status(pcr_hip_stream_synchronize(nullptr));
    }
    int64_t n = cell_count();
    for (int64_t k = 0; k < n; ++k) p[k] = value;
    return Status::success();
}

Status Grid::fill(float value) {
    for (int i = 0; i < num_bands(); ++i) {
        Status s = fill_band(i, value);
        if (!s.ok()) return s;
    }
    return Status::success();
}

Status Grid::copy_from(const Grid& other, void* stream) {
    if (!impl_ || !other.impl_) return Status::error(StatusCode::InvalidArgument, "Grid not initialized");
    if (cols() != other.cols() || rows() != other.rows() || num_bands() != other.num_bands())
        return Status::error(StatusCode::InvalidArgument, "Grid dimensions/bands mismatch");
    bool dev = false;
    for (int i = 0; i < num_bands(); ++i) {
        if (impl_->descs[i].dtype != other.impl_->descs[i].dtype)
            return Status::error(StatusCode::InvalidArgument, "Grid band dtype mismatch");
        MemoryLocation dl = impl_->bands[i].location(), sl = other.impl_->bands[i].location();
        dev = dev || dl == MemoryLocation::Device || sl == MemoryLocation::Device;
        Status s = detail::copy_bytes(impl_->bands[i].data(), dl, other.impl_->bands[i].data(), sl,
                                      impl_->bands[i].bytes(), stream);
        if (!s.ok()) return s;
    }
    if (dev && !stream) return detail::hip_status(pcr_hip_stream_synchronize(nullptr));
    return Status::success();
}

std::unique_ptr<Grid> Grid::to(MemoryLocation dst) const {
    if (!impl_) return nullptr;
    auto g = create(cols(), rows(), impl_->descs, dst);
    if (!g || !g->copy_from(*this, nullptr).ok()) return nullptr;
    return g;
}

std::unique_ptr<Grid> Grid::to_device_async(void* stream) const {
    if (!impl_) return nullptr;
    auto g = create(cols(), rows(), impl_->descs, MemoryLocation::Device);
    if (!g || !g->copy_from(*this, stream).ok()) return nullptr;
    return g;
}

std::vector<uint8_t> Grid::valid_mask(int band) const {
    std::vector<uint8_t> mask;
    const float* p = band_f32(band);
    if (!p || impl_->bands[band].location() == MemoryLocation::Device) return mask;
    int64_t n = cell_count();
    mask.resize(static_cast<size_t>(n));
    for (int64_t k = 0; k < n; ++k) mask[k] = std::isnan(p[k]) ? 0 : 1;
    return mask;
}

}  // namespace pcr
