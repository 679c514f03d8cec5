// comm.hip -- pcr_hip_comm_*: the row-block shards' exchange step over RCCL (xGMI), native side.
//
// One process per GPU; rank r owns rows [own_row0, own_row1) of the grid and keeps `halo` apron rows on each side
// (SURVEY section 8e).  The only data-path exchange is the NEIGHBOUR HALO REDUCE: a rank's apron rows go to the
// rank that owns them (ncclSend / ncclRecv to rank +- 1 inside one ncclGroup, point-to-point over one xGMI link),
// the owner merges them with the plane's op (add for sum / weight planes, max / min otherwise), plus one MAX
// all-reduce of the touched-tile flags (one word per reference tile).  Never a full-grid collective.
// Everything is enqueued on the caller's stream: ordered after the scatter kernels and before the finalize kernels
// by stream order alone.
//
// RCCL is resolved at run time (dlopen of librccl.so.1: the copy a host process -- e.g. torch -- has already loaded
// is reused, otherwise ROCm's): a single-GPU user of libpcr_hip.so never needs it.  The communicator is
// bootstrapped from a caller-supplied ncclUniqueId (128 bytes, pcr_hip_comm_unique_id on rank 0, carried to the other
// ranks by whatever the host has: MPI, a file, torch.distributed).
//
// The reference is single-device (cuda_device_id, include/pcr/engine/pipeline.h:68): nothing replaced, new work.
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

using namespace pcrhip;

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // ONE RCCL per process, and the one that belongs to the HIP runtime in use:
        //  1. a copy the process has already loaded (torch's bundled librccl.so has no SONAME: it is known by that name);
        //  2. PCR_HIP_RCCL = explicit path;
        //  3. the sibling of the libamdhip64 this library is running on (torch/lib/librccl.so next to torch's runtime,
        //     /opt/rocm/lib/librccl.so.1 next to ROCm's) -- a second RCCL with its own rocm_smi / roctx copies next to
        //     the host's ends in a double free at process exit;
        //  4. the loader's search path.
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            if (const char* forced = std::getenv("PCR_HIP_RCCL")) r.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) {
            Dl_info info;
            if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
                std::string dir(info.dli_fname);
                const size_t slash = dir.rfind('/');
                if (slash != std::string::npos) {
                    dir.resize(slash + 1);
                    for (const char* leaf : {"librccl.so.1", "librccl.so"}) {
                        r.lib = dlopen((dir + leaf).c_str(), RTLD_NOW | RTLD_GLOBAL);
                        if (r.lib) break;
                    }
                }
            }
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) { r.error = std::string("RCCL not found: ") + dlerror(); return; }
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(r.lib, n);
            if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

int rccl_fail(Rccl* r, ncclResult_t res, const char* what) {
    return fail(PCR_HIP_CUDA_ERROR, std::string("RCCL error in ") + what + ": " +
                                        (r->GetErrorString ? r->GetErrorString(res) : "unknown"));
}

#define PCR_RCCL_TRY(r, call, what)                                 \
    do {                                                            \
        ncclResult_t pcr_res_ = (call);                             \
        if (pcr_res_ != ncclSuccess) return rccl_fail(r, pcr_res_, what); \
    } while (0)

// dst[i] = op(dst[i], src[i]) for the plane's kind
__global__ void __launch_bounds__(256)
k_halo_merge(uint32_t kind, float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float a = dst[i], b = src[i];
        float o;
        if (kind == PCR_HIP_PLANE_MAX) o = fmaxf(a, b);
        else if (kind == PCR_HIP_PLANE_MIN) o = fminf(a, b);
        else o = a + b;
        dst[i] = o;
    }
}

}  // namespace

struct pcr_hip_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    float* d_recv = nullptr;            // grow-only landing area for the neighbours' rows
    size_t recv_cap = 0;
    uint64_t halo_reduces = 0, bytes_sent = 0;
};

extern "C" {

int pcr_hip_comm_available(void) {
    Rccl* r = rccl();
    return r->lib && r->error.empty() ? 1 : 0;
}

int pcr_hip_comm_unique_id(uint8_t* id128) {
    PCR_REQUIRE(id128, "comm_unique_id: null buffer");
    Rccl* r = rccl();
    if (!r->error.empty()) return fail(PCR_HIP_NOT_IMPLEMENTED, r->error);
    static_assert(sizeof(ncclUniqueId) == PCR_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    PCR_RCCL_TRY(r, r->GetUniqueId(&id), "ncclGetUniqueId");
    __builtin_memcpy(id128, &id, sizeof id);
    return PCR_HIP_OK;
}

int pcr_hip_comm_create(pcr_hip_comm** out, const uint8_t* id128, int rank, int world, int device) {
    PCR_REQUIRE(out && id128, "comm_create: null argument");
    *out = nullptr;
    PCR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_create: rank outside [0, world)");
    Rccl* r = rccl();
    if (!r->error.empty()) return fail(PCR_HIP_NOT_IMPLEMENTED, r->error);
    int prev = -1;
    PCR_HIP_TRY(hipGetDevice(&prev));
    PCR_HIP_TRY(hipSetDevice(device));
    auto* c = new (std::nothrow) pcr_hip_comm();
    if (!c) { (void)hipSetDevice(prev); return fail(PCR_HIP_OUT_OF_MEMORY, "comm_create: host allocation failed"); }
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId id;
    __builtin_memcpy(&id, id128, sizeof id);
    ncclResult_t res = r->CommInitRank(&c->comm, world, id, rank);
    (void)hipSetDevice(prev);
    if (res != ncclSuccess) { delete c; return rccl_fail(r, res, "ncclCommInitRank"); }
    *out = c;
    return PCR_HIP_OK;
}

int pcr_hip_comm_destroy(pcr_hip_comm* c) {
    if (!c) return PCR_HIP_OK;
    Rccl* r = rccl();
    if (c->comm && r->CommDestroy) (void)r->CommDestroy(c->comm);
    if (c->d_recv) (void)hipFree(c->d_recv);
    delete c;
    return PCR_HIP_OK;
}

int pcr_hip_comm_rank(const pcr_hip_comm* c, int* rank, int* world) {
    PCR_REQUIRE(c, "comm_rank: null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return PCR_HIP_OK;
}

int pcr_hip_comm_halo_reduce(pcr_hip_comm* c, const pcr_hip_halo_plane* planes, int nplanes, int width,
                             int state_row0, int state_rows, int own_row0, int own_row1, int halo, pcr_hip_stream s) {
    PCR_REQUIRE(c, "comm_halo_reduce: null communicator");
    PCR_REQUIRE(nplanes >= 0 && (nplanes == 0 || planes), "comm_halo_reduce: null planes");
    PCR_REQUIRE(width > 0 && state_rows > 0 && own_row0 >= state_row0 && own_row1 > own_row0 &&
                    own_row1 <= state_row0 + state_rows && halo >= 0,
                "comm_halo_reduce: the owned rows must lie inside the state window");
    if (c->world == 1 || halo == 0 || nplanes == 0) return PCR_HIP_OK;
    // a footprint reaches at most `halo` rows: only rank +- 1 hold rows of mine as long as every block is that tall
    PCR_REQUIRE(own_row1 - own_row0 >= halo, "comm_halo_reduce: row block shorter than the glyph halo: use fewer ranks or a smaller radius");
    Rccl* r = rccl();
    hipStream_t st = static_cast<hipStream_t>(s);
    const int up_n = own_row0 - state_row0;                       // apron rows I hold above my block: they belong to rank - 1
    const int dn_n = state_row0 + state_rows - own_row1;          // ... below: rank + 1
    const int own_n = own_row1 - own_row0;
    const int recv_n = halo < own_n ? halo : own_n;               // a neighbour holds min(halo, my rows) of my rows
    const bool has_up = c->rank > 0, has_dn = c->rank < c->world - 1;
    const size_t slot = (size_t)recv_n * width;                   // floats per (plane, neighbour)
    const size_t need = slot * 2 * (size_t)nplanes;
    int prev = -1;
    PCR_HIP_TRY(hipGetDevice(&prev));
    if (prev != c->device) PCR_HIP_TRY(hipSetDevice(c->device));
    struct Restore { int prev, dev; ~Restore() { if (prev != dev) (void)hipSetDevice(prev); } } restore{prev, c->device};
    if (need > c->recv_cap) {
        PCR_HIP_TRY(hipStreamSynchronize(st));                    // earlier merges may still read the old block
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_recv = nullptr;
        c->recv_cap = 0;
        PCR_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_recv), need * sizeof(float)));
        c->recv_cap = need;
    }
    PCR_RCCL_TRY(r, r->GroupStart(), "ncclGroupStart");
    for (int p = 0; p < nplanes; ++p) {
        float* plane = planes[p].d_plane;
        if (has_up) {
            if (up_n > 0) {
                PCR_RCCL_TRY(r, r->Send(plane, (size_t)up_n * width, ncclFloat, c->rank - 1, c->comm, st), "ncclSend");
                c->bytes_sent += (uint64_t)up_n * width * 4;
            }
            PCR_RCCL_TRY(r, r->Recv(c->d_recv + (size_t)(2 * p) * slot, slot, ncclFloat, c->rank - 1, c->comm, st), "ncclRecv");
        }
        if (has_dn) {
            if (dn_n > 0) {
                PCR_RCCL_TRY(r, r->Send(plane + (size_t)(state_rows - dn_n) * width, (size_t)dn_n * width, ncclFloat,
                                        c->rank + 1, c->comm, st), "ncclSend");
                c->bytes_sent += (uint64_t)dn_n * width * 4;
            }
            PCR_RCCL_TRY(r, r->Recv(c->d_recv + (size_t)(2 * p + 1) * slot, slot, ncclFloat, c->rank + 1, c->comm, st), "ncclRecv");
        }
    }
    PCR_RCCL_TRY(r, r->GroupEnd(), "ncclGroupEnd");
    // what rank - 1 sent is its bottom apron = my first rows; what rank + 1 sent is its top apron = my last rows
    const int blocks = (int)((slot + 255) / 256 < 2048 ? (slot + 255) / 256 : 2048);
    for (int p = 0; p < nplanes; ++p) {
        float* plane = planes[p].d_plane;
        if (has_up)
            hipLaunchKernelGGL(k_halo_merge, dim3(blocks), dim3(256), 0, st, planes[p].kind,
                               plane + (size_t)up_n * width, c->d_recv + (size_t)(2 * p) * slot, (int64_t)slot);
        if (has_dn)
            hipLaunchKernelGGL(k_halo_merge, dim3(blocks), dim3(256), 0, st, planes[p].kind,
                               plane + (size_t)(up_n + own_n - recv_n) * width, c->d_recv + (size_t)(2 * p + 1) * slot, (int64_t)slot);
    }
    PCR_HIP_TRY(hipGetLastError());
    c->halo_reduces++;
    return PCR_HIP_OK;
}

int pcr_hip_comm_allreduce_max_u32(pcr_hip_comm* c, uint32_t* d_words, int count, pcr_hip_stream s) {
    PCR_REQUIRE(c, "comm_allreduce_max_u32: null communicator");
    PCR_REQUIRE(count >= 0 && (count == 0 || d_words), "comm_allreduce_max_u32: null words");
    if (c->world == 1 || count == 0) return PCR_HIP_OK;
    Rccl* r = rccl();
    PCR_RCCL_TRY(r, r->AllReduce(d_words, d_words, (size_t)count, ncclUint32, ncclMax, c->comm, static_cast<hipStream_t>(s)),
                 "ncclAllReduce");
    return PCR_HIP_OK;
}

int pcr_hip_comm_allreduce_sum_f64(pcr_hip_comm* c, double* d_values, int count, pcr_hip_stream s) {
    PCR_REQUIRE(c, "comm_allreduce_sum_f64: null communicator");
    PCR_REQUIRE(count >= 0 && (count == 0 || d_values), "comm_allreduce_sum_f64: null values");
    if (c->world == 1 || count == 0) return PCR_HIP_OK;
    Rccl* r = rccl();
    PCR_RCCL_TRY(r, r->AllReduce(d_values, d_values, (size_t)count, ncclFloat64, ncclSum, c->comm, static_cast<hipStream_t>(s)),
                 "ncclAllReduce");
    return PCR_HIP_OK;
}

int pcr_hip_comm_stats(const pcr_hip_comm* c, uint64_t* halo_reduces, uint64_t* bytes_sent) {
    PCR_REQUIRE(c, "comm_stats: null communicator");
    if (halo_reduces) *halo_reduces = c->halo_reduces;
    if (bytes_sent) *bytes_sent = c->bytes_sent;
    return PCR_HIP_OK;
}

}  // extern "C"
