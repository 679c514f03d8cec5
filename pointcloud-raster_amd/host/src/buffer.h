// buffer.h -- (internal) one allocation in Host, pinned-host or Device memory; device and
// pinned storage go through the HIP C-ABI.
#pragma once

#include "pcr/core/types.h"
#include "pcr_hip.h"

#include <cstdlib>
#include <cstring>

namespace pcr {
namespace detail {

inline Status hip_status(int rc) {
    if (rc == PCR_HIP_OK) return Status::success();
    return Status::error(static_cast<StatusCode>(rc), pcr_hip_last_error());
}

class Buffer {
public:
    Buffer() = default;
    Buffer(const Buffer&) = delete;
    Buffer& operator=(const Buffer&) = delete;
    Buffer(Buffer&& o) noexcept { steal(o); }
    Buffer& operator=(Buffer&& o) noexcept {
        if (this != &o) { release(); steal(o); }
        return *this;
    }
    ~Buffer() { release(); }

    Status allocate(size_t bytes, MemoryLocation loc) {
        release();
        loc_ = loc;
        bytes_ = bytes;
        size_t want = bytes ? bytes : 1;
        switch (loc) {
            case MemoryLocation::Host:
                ptr_ = std::malloc(want);
                if (!ptr_) return Status::error(StatusCode::OutOfMemory, "host allocation failed");
                return Status::success();
            case MemoryLocation::HostPinned:
                return hip_status(pcr_hip_host_alloc(&ptr_, want));
            case MemoryLocation::Device:
                return hip_status(pcr_hip_malloc(&ptr_, want));
        }
        return Status::error(StatusCode::InvalidArgument, "unknown memory location");
    }

    void release() {
        if (!ptr_) return;
        switch (loc_) {
            case MemoryLocation::Host: std::free(ptr_); break;
            case MemoryLocation::HostPinned: pcr_hip_host_free(ptr_); break;
            case MemoryLocation::Device: pcr_hip_free(ptr_); break;
        }
        ptr_ = nullptr;
        bytes_ = 0;
    }

    void* data() const { return ptr_; }
    size_t bytes() const { return bytes_; }
    MemoryLocation location() const { return loc_; }

private:
    void steal(Buffer& o) {
        ptr_ = o.ptr_; bytes_ = o.bytes_; loc_ = o.loc_;
        o.ptr_ = nullptr; o.bytes_ = 0;
    }
    void* ptr_ = nullptr;
    size_t bytes_ = 0;
    MemoryLocation loc_ = MemoryLocation::Host;
};

// Copy between any two locations; async on `stream` where the runtime allows it.
inline Status copy_bytes(void* dst, MemoryLocation dloc, const void* src, MemoryLocation sloc,
                         size_t bytes, void* stream) {
    if (bytes == 0) return Status::success();
    bool ddev = dloc == MemoryLocation::Device, sdev = sloc == MemoryLocation::Device;
    if (!ddev && !sdev) { std::memcpy(dst, src, bytes); return Status::success(); }
    if (ddev && sdev) return hip_status(pcr_hip_memcpy_d2d(dst, src, bytes, stream));
    if (ddev) return hip_status(pcr_hip_memcpy_h2d(dst, src, bytes, stream));
    return hip_status(pcr_hip_memcpy_d2h(dst, src, bytes, stream));
}

}  // namespace detail
}  // namespace pcr
