// point_cloud.cpp -- SoA point cloud storage.  Contract: the reference's
// src/core/point_cloud.cpp (create(0) -> nullptr :209-211, channels inherit the cloud's
// location, resize never reallocates, to() deep-copies coordinates and every channel).
#include "pcr/core/point_cloud.h"

#include "buffer.h"

#include <map>

namespace pcr {

struct PointCloud::Impl {
    detail::Buffer xbuf, ybuf;              // owned coordinates (empty when wrapping)
    double* x = nullptr;
    double* y = nullptr;
    size_t count = 0;
    size_t capacity = 0;
    MemoryLocation loc = MemoryLocation::Host;
    CRS crs;
    std::map<std::string, ChannelDesc> descs;        // ordered: stable channel_names()
    std::map<std::string, detail::Buffer> data;
};

PointCloud::~PointCloud() = default;

std::unique_ptr<PointCloud> PointCloud::create(size_t capacity, MemoryLocation loc) {
    if (capacity == 0) return nullptr;
    auto pc = std::unique_ptr<PointCloud>(new PointCloud());
    pc->impl_ = std::make_unique<Impl>();
    Impl& m = *pc->impl_;
    m.loc = loc;
    m.capacity = capacity;
    if (!m.xbuf.allocate(capacity * sizeof(double), loc).ok()) return nullptr;
    if (!m.ybuf.allocate(capacity * sizeof(double), loc).ok()) return nullptr;
    m.x = static_cast<double*>(m.xbuf.data());
    m.y = static_cast<double*>(m.ybuf.data());
    return pc;
}

std::unique_ptr<PointCloud> PointCloud::wrap(double* x, double* y, size_t count, MemoryLocation loc) {
    if (!x || !y || count == 0) return nullptr;
    auto pc = std::unique_ptr<PointCloud>(new PointCloud());
    pc->impl_ = std::make_unique<Impl>();
    Impl& m = *pc->impl_;
    m.loc = loc;
    m.x = x;
    m.y = y;
    m.count = m.capacity = count;
    return pc;
}

Status PointCloud::add_channel(const std::string& name, DataType dtype) {
    if (!impl_) return Status::error(StatusCode::InvalidArgument, "PointCloud not initialized");
    if (impl_->descs.count(name))
        return Status::error(StatusCode::InvalidArgument, "Channel already exists: " + name);
    detail::Buffer b;
    Status s = b.allocate(impl_->capacity * data_type_size(dtype), impl_->loc);
    if (!s.ok()) return Status::error(s.code, "Failed to allocate channel: " + name);
    ChannelDesc d;
    d.name = name;
    d.dtype = dtype;
    impl_->descs[name] = d;
    impl_->data[name] = std::move(b);
    return Status::success();
}

bool PointCloud::has_channel(const std::string& name) const {
    return impl_ && impl_->descs.count(name) > 0;
}

const ChannelDesc* PointCloud::channel(const std::string& name) const {
    if (!impl_) return nullptr;
    auto it = impl_->descs.find(name);
    return it == impl_->descs.end() ? nullptr : &it->second;
}

std::vector<std::string> PointCloud::channel_names() const {
    std::vector<std::string> names;
    if (impl_) for (const auto& kv : impl_->descs) names.push_back(kv.first);
    return names;
}

double* PointCloud::x() { return impl_ ? impl_->x : nullptr; }
const double* PointCloud::x() const { return impl_ ? impl_->x : nullptr; }
double* PointCloud::y() { return impl_ ? impl_->y : nullptr; }
const double* PointCloud::y() const { return impl_ ? impl_->y : nullptr; }

void* PointCloud::channel_data(const std::string& name) {
    if (!impl_) return nullptr;
    auto it = impl_->data.find(name);
    return it == impl_->data.end() ? nullptr : it->second.data();
}

const void* PointCloud::channel_data(const std::string& name) const {
    return const_cast<PointCloud*>(this)->channel_data(name);
}

float* PointCloud::channel_f32(const std::string& name) {
    const ChannelDesc* d = channel(name);
    if (!d || d->dtype != DataType::Float32) return nullptr;
    return static_cast<float*>(channel_data(name));
}
const float* PointCloud::channel_f32(const std::string& name) const {
    return const_cast<PointCloud*>(this)->channel_f32(name);
}

int32_t* PointCloud::channel_i32(const std::string& name) {
    const ChannelDesc* d = channel(name);
    if (!d || d->dtype != DataType::Int32) return nullptr;
    return static_cast<int32_t*>(channel_data(name));
}
const int32_t* PointCloud::channel_i32(const std::string& name) const {
    return const_cast<PointCloud*>(this)->channel_i32(name);
}

size_t PointCloud::count() const { return impl_ ? impl_->count : 0; }
size_t PointCloud::capacity() const { return impl_ ? impl_->capacity : 0; }
MemoryLocation PointCloud::location() const { return impl_ ? impl_->loc : MemoryLocation::Host; }
CRS PointCloud::crs() const { return impl_ ? impl_->crs : CRS{}; }
void PointCloud::set_crs(const CRS& crs) { if (impl_) impl_->crs = crs; }

Status PointCloud::resize(size_t new_count) {
    if (!impl_) return Status::error(StatusCode::InvalidArgument, "PointCloud not initialized");
    if (new_count > impl_->capacity)
        return Status::error(StatusCode::InvalidArgument, "Cannot resize beyond capacity");
    impl_->count = new_count;
    return Status::success();
}

static std::unique_ptr<PointCloud> copy_cloud(const PointCloud& src, MemoryLocation dst, void* stream, bool sync) {
    if (src.capacity() == 0) return nullptr;
    auto out = PointCloud::create(src.capacity(), dst);
    if (!out) return nullptr;
    out->resize(src.count());
    out->set_crs(src.crs());
    size_t n = src.count();
    MemoryLocation sl = src.location();
    if (!detail::copy_bytes(out->x(), dst, src.x(), sl, n * sizeof(double), stream).ok()) return nullptr;
    if (!detail::copy_bytes(out->y(), dst, src.y(), sl, n * sizeof(double), stream).ok()) return nullptr;
    for (const std::string& name : src.channel_names()) {
        const ChannelDesc* d = src.channel(name);
        if (!out->add_channel(name, d->dtype).ok()) return nullptr;
        if (!detail::copy_bytes(out->channel_data(name), dst, src.channel_data(name), sl,
                                n * data_type_size(d->dtype), stream).ok())
            return nullptr;
    }
    bool touches_device = dst == MemoryLocation::Device || sl == MemoryLocation::Device;
    if (sync && touches_device && pcr_hip_stream_synchronize(stream) != PCR_HIP_OK) return nullptr;
    return out;
}

std::unique_ptr<PointCloud> PointCloud::to(MemoryLocation dst) const {
    if (!impl_) return nullptr;
    return copy_cloud(*this, dst, nullptr, true);
}

std::unique_ptr<PointCloud> PointCloud::to_device_async(void* stream) const {
    if (!impl_) return nullptr;
    return copy_cloud(*this, MemoryLocation::Device, stream, false);
}

}  // namespace pcr
