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

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
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
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // ONE RCCL per process, and the one that belongs to the HIP runtime in use:
        //  1. PCR_HIP_RCCL = explicit path: what the caller names wins (a site-specific build; the test double of
        //     tests/native/fake_rccl.cpp, which lets several ranks share one GPU);
        //  2. a copy the process has already loaded (torch's bundled librccl.so has no SONAME: it is known by that name);
        //  3. the sibling of the libamdhip64 this library is running on (torch/lib/librccl.so next to torch's runtime,
        //     /opt/rocm/lib/librccl.so.1 next to ROCm's) -- a second RCCL with its own rocm_smi / roctx copies next to
        //     the host's ends in a double free at process exit;
        //  4. the loader's search path.
        if (const char* forced = std::getenv("PCR_HIP_RCCL")) {
            if (forced[0]) r.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        }
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
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
        // TWO copies of RCCL in one process (a host application that loaded ROCm's librccl before it imported torch, whose
        // wheel bundles its own) each bring their rocm_smi / roctx statics: round 3 saw `double free or corruption` at
        // process exit.  The mapping table says whether that is the case; refuse to add communicators on top of it.
        {
            std::set<std::string> copies;
            if (FILE* maps = std::fopen("/proc/self/maps", "r")) {
                char line[1024];
                while (std::fgets(line, sizeof line, maps)) {
                    const char* path = std::strchr(line, '/');
                    if (!path) continue;
                    const char* leaf = std::strrchr(path, '/');
                    if (leaf && std::strncmp(leaf + 1, "librccl", 7) == 0) {
                        std::string full(path);
                        while (!full.empty() && (full.back() == '\n' || full.back() == ' ')) full.pop_back();
                        copies.insert(full);
                    }
                }
                std::fclose(maps);
            }
            if (copies.size() > 1) {
                r.error = "two copies of RCCL are mapped into this process (";
                for (const auto& c : copies) r.error += c + "; ";
                r.error += "): load one RCCL only -- e.g. import torch before anything that loads ROCm's librccl, or set PCR_HIP_RCCL";
                return;
            }
        }
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
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
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

// dst[i] = identity of the plane's kind: what an apron row holds once its contents have gone to the row's owner
__global__ void __launch_bounds__(256)
k_halo_reset(uint32_t kind, float* __restrict__ dst, int64_t n) {
    const float id = kind == PCR_HIP_PLANE_MAX ? -FLT_MAX : kind == PCR_HIP_PLANE_MIN ? FLT_MAX : 0.0f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = id;
}

}  // namespace

struct pcr_hip_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    float* d_recv = nullptr;            // grow-only landing area for the neighbours' rows
    size_t recv_cap = 0;
    int32_t* d_agree = nullptr;         // (world + 1) agreement records: [0] = mine, [1 ..] = everyone's (all-gather target)
    int32_t* h_agree = nullptr;         // page-locked mirror
    size_t agree_ints = 0;              // int32 per record the two blocks are sized for
    uint64_t halo_reduces = 0, bytes_sent = 0, agreements = 0;
};

namespace {

constexpr int kGeomInts = (int)(sizeof(pcr_hip_halo_geom) / sizeof(int32_t));
static_assert(sizeof(pcr_hip_halo_geom) == 10 * sizeof(int32_t), "pcr_hip_halo_geom is ten int32");

// Every rank's record to every rank: H2D of mine, ncclAllGather, D2H of all, stream sync.  The ONLY thing a rank does
// before it knows what the others brought -- so it is posted unconditionally, whatever this rank's own arguments were.
int ensure_agree(pcr_hip_comm* c, size_t ints, hipStream_t st) {
    if (c->d_agree && c->agree_ints >= ints) return PCR_HIP_OK;
    if (c->d_agree) {
        PCR_HIP_TRY(hipStreamSynchronize(st));                    // (an earlier agreement's copies may still be in flight)
        (void)hipFree(c->d_agree);
        (void)hipHostFree(c->h_agree);
        c->d_agree = c->h_agree = nullptr;
        c->agree_ints = 0;
    }
    PCR_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_agree), (size_t)(c->world + 1) * ints * sizeof(int32_t)));
    PCR_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_agree), (size_t)(c->world + 1) * ints * sizeof(int32_t), hipHostMallocDefault));
    c->agree_ints = ints;
    return PCR_HIP_OK;
}

// One record of `ints` int32 from every rank to every rank.
int gather_records(pcr_hip_comm* c, Rccl* r, const void* mine, void* all, size_t ints, hipStream_t st) {
    int rc = ensure_agree(c, ints, st);
    if (rc) return rc;
    __builtin_memcpy(c->h_agree, mine, ints * 4);
    PCR_HIP_TRY(hipMemcpyAsync(c->d_agree, c->h_agree, ints * 4, hipMemcpyHostToDevice, st));
    PCR_RCCL_TRY(r, r->AllGather(c->d_agree, c->d_agree + ints, ints, ncclInt32, c->comm, st), "ncclAllGather");
    PCR_HIP_TRY(hipMemcpyAsync(c->h_agree + ints, c->d_agree + ints, (size_t)c->world * ints * 4, hipMemcpyDeviceToHost, st));
    PCR_HIP_TRY(hipStreamSynchronize(st));
    __builtin_memcpy(all, c->h_agree + ints, (size_t)c->world * ints * 4);
    c->agreements++;
    return PCR_HIP_OK;
}

int gather_geoms(pcr_hip_comm* c, Rccl* r, const pcr_hip_halo_geom& mine, pcr_hip_halo_geom* all, hipStream_t st) {
    return gather_records(c, r, &mine, all, (size_t)kGeomInts, st);
}

}  // namespace

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
    if (c->d_agree) (void)hipFree(c->d_agree);
    if (c->h_agree) (void)hipHostFree(c->h_agree);
    delete c;
    return PCR_HIP_OK;
}

int pcr_hip_comm_rank(const pcr_hip_comm* c, int* rank, int* world) {
    PCR_REQUIRE(c, "comm_rank: null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return PCR_HIP_OK;
}

// The judgement every rank passes on the gathered geometries: a pure function of `all`, so that every rank reaches the
// SAME verdict -- either all post their sends and receives (with sizes that match by construction: a receive is sized
// from what the sender says it holds), or all return PCR_HIP_INVALID_ARGUMENT and nobody waits for anybody.
int pcr_hip_comm_halo_plan(const pcr_hip_halo_geom* all, int world, int rank, int* send_up_rows, int* send_dn_rows,
                           int* recv_up_rows, int* recv_dn_rows) {
    PCR_REQUIRE(all && world >= 1 && rank >= 0 && rank < world, "comm_halo_plan: null geometries or rank outside [0, world)");
    PCR_REQUIRE(send_up_rows && send_dn_rows && recv_up_rows && recv_dn_rows, "comm_halo_plan: null result pointer");
    *send_up_rows = *send_dn_rows = *recv_up_rows = *recv_dn_rows = 0;
    auto who = [](int r) { return "rank " + std::to_string(r); };
    for (int r = 0; r < world; ++r) {
        const pcr_hip_halo_geom& g = all[r];
        if (!g.valid) return fail(PCR_HIP_INVALID_ARGUMENT, "comm_halo_reduce: " + who(r) + " was called with invalid arguments (null planes, "
                                  "non-positive width or state window) or could not allocate its landing area; refused on every rank");
        if (g.width != all[0].width || g.halo != all[0].halo || g.nplanes != all[0].nplanes || g.kinds != all[0].kinds)
            return fail(PCR_HIP_INVALID_ARGUMENT, "comm_halo_reduce: " + who(r) + " brings width / halo / planes " +
                        std::to_string(g.width) + " / " + std::to_string(g.halo) + " / " + std::to_string(g.nplanes) + " where rank 0 brings " +
                        std::to_string(all[0].width) + " / " + std::to_string(all[0].halo) + " / " + std::to_string(all[0].nplanes) +
                        "; refused on every rank");
        if (!(g.own_row1 > g.own_row0)) return fail(PCR_HIP_INVALID_ARGUMENT, "comm_halo_reduce: " + who(r) + " owns no rows: use fewer ranks");
        if (!(g.own_row0 >= g.state_row0 && g.own_row1 <= g.state_row0 + g.state_rows))
            return fail(PCR_HIP_INVALID_ARGUMENT, "comm_halo_reduce: the owned rows of " + who(r) + " must lie inside its state window");
        if (r > 0 && all[r - 1].own_row1 != g.own_row0)
            return fail(PCR_HIP_INVALID_ARGUMENT, "comm_halo_reduce: the row blocks of " + who(r - 1) + " and " + who(r) + " are not contiguous");
    }
    for (int r = 0; r < world; ++r) {
        const pcr_hip_halo_geom& g = all[r];
        const int up = g.own_row0 - g.state_row0, dn = g.state_row0 + g.state_rows - g.own_row1;
        if (up > g.halo || dn > g.halo)
            return fail(PCR_HIP_INVALID_ARGUMENT, "comm_halo_reduce: " + who(r) + " holds " + std::to_string(up > dn ? up : dn) +
                        " apron rows beside its block, more than the halo of " + std::to_string(g.halo));
        // the apron rows a rank holds beyond its block go to ONE neighbour: they must all be that neighbour's rows
        if (r > 0 && up > all[r - 1].own_row1 - all[r - 1].own_row0)
            return fail(PCR_HIP_INVALID_ARGUMENT, "comm_halo_reduce: the row block of " + who(r - 1) + " (" +
                        std::to_string(all[r - 1].own_row1 - all[r - 1].own_row0) + " rows) is shorter than the " + std::to_string(up) +
                        " apron rows " + who(r) + " holds above its block: use fewer ranks or a smaller radius");
        if (r < world - 1 && dn > all[r + 1].own_row1 - all[r + 1].own_row0)
            return fail(PCR_HIP_INVALID_ARGUMENT, "comm_halo_reduce: the row block of " + who(r + 1) + " (" +
                        std::to_string(all[r + 1].own_row1 - all[r + 1].own_row0) + " rows) is shorter than the " + std::to_string(dn) +
                        " apron rows " + who(r) + " holds below its block: use fewer ranks or a smaller radius");
    }
    if (all[0].halo == 0 || all[0].nplanes == 0) return PCR_HIP_OK;          // agreed: nothing to move
    const pcr_hip_halo_geom& me = all[rank];
    if (rank > 0) {
        *send_up_rows = me.own_row0 - me.state_row0;
        const pcr_hip_halo_geom& nb = all[rank - 1];
        *recv_up_rows = nb.state_row0 + nb.state_rows - nb.own_row1;           // its bottom apron = my first rows
    }
    if (rank < world - 1) {
        *send_dn_rows = me.state_row0 + me.state_rows - me.own_row1;
        const pcr_hip_halo_geom& nb = all[rank + 1];
        *recv_dn_rows = nb.own_row0 - nb.state_row0;                           // its top apron = my last rows
    }
    return PCR_HIP_OK;
}

int pcr_hip_comm_halo_reduce(pcr_hip_comm* c, const pcr_hip_halo_plane* planes, int nplanes, int width,
                             int state_row0, int state_rows, int own_row0, int own_row1, int halo, pcr_hip_stream s) {
    PCR_REQUIRE(c, "comm_halo_reduce: null communicator");
    hipStream_t st = static_cast<hipStream_t>(s);
    // What this rank brings.  A rank whose own arguments are bad does NOT return yet: it says so in its record, takes
    // part in the agreement like everybody else, and every rank refuses together (one rank returning alone would leave
    // its neighbours in ncclRecv for ever).
    pcr_hip_halo_geom mine{};
    mine.width = width; mine.state_row0 = state_row0; mine.state_rows = state_rows;
    mine.own_row0 = own_row0; mine.own_row1 = own_row1; mine.halo = halo; mine.nplanes = nplanes;
    bool ok_local = nplanes >= 0 && nplanes <= 8 && (nplanes == 0 || planes) && width > 0 && state_rows > 0 && halo >= 0;
    uint32_t kinds = 0;
    for (int p = 0; ok_local && p < nplanes; ++p) {
        ok_local = planes[p].d_plane != nullptr && planes[p].kind != 0 && planes[p].kind < 16;
        kinds |= (planes[p].kind & 15u) << (4 * p);
    }
    mine.kinds = (int32_t)kinds;
    mine.valid = ok_local ? 1 : 0;
    if (c->world == 1) {
        PCR_REQUIRE(ok_local && own_row0 >= state_row0 && own_row1 > own_row0 && own_row1 <= state_row0 + state_rows,
                    "comm_halo_reduce: the owned rows must lie inside the state window");
        return PCR_HIP_OK;
    }
    Rccl* r = rccl();
    int prev = -1;
    PCR_HIP_TRY(hipGetDevice(&prev));
    if (prev != c->device) PCR_HIP_TRY(hipSetDevice(c->device));
    struct Restore { int prev, dev; ~Restore() { if (prev != dev) (void)hipSetDevice(prev); } } restore{prev, c->device};

    // The landing area is sized BEFORE the agreement, from this rank's own arguments (an apron is at most `halo` rows: the
    // plan refuses anything larger), so that an allocation failure is part of this rank's record too.
    if (ok_local && halo > 0 && nplanes > 0) {
        const size_t need = 2 * (size_t)halo * (size_t)width * (size_t)nplanes;
        if (need > c->recv_cap) {
            bool grown = hipStreamSynchronize(st) == hipSuccess;      // earlier merges may still read the old block
            if (grown) {
                if (c->d_recv) (void)hipFree(c->d_recv);
                c->d_recv = nullptr;
                c->recv_cap = 0;
                grown = hipMalloc(reinterpret_cast<void**>(&c->d_recv), need * sizeof(float)) == hipSuccess;
            }
            if (grown) c->recv_cap = need;
            else { (void)hipGetLastError(); mine.valid = 0; }
        }
    }
    std::vector<pcr_hip_halo_geom> all((size_t)c->world);
    int rc = gather_geoms(c, r, mine, all.data(), st);
    if (rc) return rc;
    int send_up = 0, send_dn = 0, recv_up = 0, recv_dn = 0;
    rc = pcr_hip_comm_halo_plan(all.data(), c->world, c->rank, &send_up, &send_dn, &recv_up, &recv_dn);
    if (rc) return rc;                                            // the same verdict on every rank
    if (halo == 0 || nplanes == 0) return PCR_HIP_OK;             // (agreed above: the same on every rank)

    const size_t up_slot = (size_t)recv_up * width, dn_slot = (size_t)recv_dn * width;     // floats per plane (<= halo * width each)
    const int own_n = own_row1 - own_row0, up_n = own_row0 - state_row0;
    // Everything between GroupStart and GroupEnd is posted even after a failure: a group left open would swallow the
    // thread's next RCCL call.  The first error is reported after the group is closed.
    ncclResult_t first = ncclSuccess;
    const char* where = "";
    auto note = [&](ncclResult_t res, const char* what) { if (res != ncclSuccess && first == ncclSuccess) { first = res; where = what; } };
    PCR_RCCL_TRY(r, r->GroupStart(), "ncclGroupStart");
    for (int p = 0; p < nplanes; ++p) {
        float* plane = planes[p].d_plane;
        float* land = c->d_recv + (size_t)p * (up_slot + dn_slot);
        if (send_up > 0) {
            note(r->Send(plane, (size_t)send_up * width, ncclFloat, c->rank - 1, c->comm, st), "ncclSend");
            c->bytes_sent += (uint64_t)send_up * width * 4;
        }
        if (recv_up > 0) note(r->Recv(land, up_slot, ncclFloat, c->rank - 1, c->comm, st), "ncclRecv");
        if (send_dn > 0) {
            note(r->Send(plane + (size_t)(state_rows - send_dn) * width, (size_t)send_dn * width, ncclFloat, c->rank + 1, c->comm, st), "ncclSend");
            c->bytes_sent += (uint64_t)send_dn * width * 4;
        }
        if (recv_dn > 0) note(r->Recv(land + up_slot, dn_slot, ncclFloat, c->rank + 1, c->comm, st), "ncclRecv");
    }
    note(r->GroupEnd(), "ncclGroupEnd");
    if (first != ncclSuccess) return rccl_fail(r, first, where);
    // what rank - 1 sent is its bottom apron = my first rows; what rank + 1 sent is its top apron = my last rows
    auto grid_of = [](size_t n) { return (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048); };
    for (int p = 0; p < nplanes; ++p) {
        float* plane = planes[p].d_plane;
        const float* land = c->d_recv + (size_t)p * (up_slot + dn_slot);
        // The apron rows that were sent now live in their owner's rows: back to the plane's identity (stream order puts
        // this after the sends), so that a LATER exchange -- the reference keeps state across finalize and its benchmarks
        // re-finalize one pipeline, src/engine/pipeline.cpp:1344-1364 -- carries only what was accumulated since.
        if (send_up > 0)
            hipLaunchKernelGGL(k_halo_reset, dim3(grid_of((size_t)send_up * width)), dim3(256), 0, st, planes[p].kind,
                               plane, (int64_t)send_up * width);
        if (send_dn > 0)
            hipLaunchKernelGGL(k_halo_reset, dim3(grid_of((size_t)send_dn * width)), dim3(256), 0, st, planes[p].kind,
                               plane + (size_t)(state_rows - send_dn) * width, (int64_t)send_dn * width);
        if (recv_up > 0)
            hipLaunchKernelGGL(k_halo_merge, dim3(grid_of(up_slot)), dim3(256), 0, st, planes[p].kind,
                               plane + (size_t)up_n * width, land, (int64_t)up_slot);
        if (recv_dn > 0)
            hipLaunchKernelGGL(k_halo_merge, dim3(grid_of(dn_slot)), dim3(256), 0, st, planes[p].kind,
                               plane + (size_t)(up_n + own_n - recv_dn) * width, land + up_slot, (int64_t)dn_slot);
    }
    PCR_HIP_TRY(hipGetLastError());
    c->halo_reduces++;
    return PCR_HIP_OK;
}

// MAX over the ranks of one host integer (H2D, ncclAllReduce, D2H, stream sync): what a sharded ingest agrees on before
// anything is accumulated -- e.g. the Line reach of this round's clouds -- so that every rank refuses together.
int pcr_hip_comm_agree_max_i32(pcr_hip_comm* c, int32_t* h_inout, pcr_hip_stream s) {
    PCR_REQUIRE(c && h_inout, "comm_agree_max_i32: null argument");
    if (c->world == 1) return PCR_HIP_OK;
    Rccl* r = rccl();
    hipStream_t st = static_cast<hipStream_t>(s);
    int prev = -1;
    PCR_HIP_TRY(hipGetDevice(&prev));
    if (prev != c->device) PCR_HIP_TRY(hipSetDevice(c->device));
    struct Restore { int prev, dev; ~Restore() { if (prev != dev) (void)hipSetDevice(prev); } } restore{prev, c->device};
    { int rc = ensure_agree(c, (size_t)kGeomInts, st); if (rc) return rc; }
    c->h_agree[0] = *h_inout;
    PCR_HIP_TRY(hipMemcpyAsync(c->d_agree, c->h_agree, sizeof(int32_t), hipMemcpyHostToDevice, st));
    PCR_RCCL_TRY(r, r->AllReduce(c->d_agree, c->d_agree, 1, ncclInt32, ncclMax, c->comm, st), "ncclAllReduce");
    PCR_HIP_TRY(hipMemcpyAsync(c->h_agree, c->d_agree, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PCR_HIP_TRY(hipStreamSynchronize(st));
    *h_inout = c->h_agree[0];
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

// ---- variable-size transfers between the ranks: all-to-all (an unrouted cloud's points to their owners) and gather
//      (the ranks' finished row strips to one rank), with the same discipline as the halo reduce: what every rank brings is
//      all-gathered first, every rank passes the same verdict on it, and a receive is sized from what its sender announced.

namespace {
constexpr int kXferInts = (int)(sizeof(pcr_hip_xfer_geom) / sizeof(int32_t));
static_assert(sizeof(pcr_hip_xfer_geom) % sizeof(int32_t) == 0, "pcr_hip_xfer_geom is whole int32 words");

int xfer(pcr_hip_comm* c, int narrays, const void* const* d_send, void* const* d_recv, const int32_t* elem_bytes,
         const uint64_t* h_send_counts, uint64_t recv_capacity, uint64_t* h_recv_counts, int root, pcr_hip_stream s, const char* what) {
    PCR_REQUIRE(c, std::string(what) + ": null communicator");
    hipStream_t st = static_cast<hipStream_t>(s);
    pcr_hip_xfer_geom mine{};
    mine.narrays = narrays;
    mine.root = root;
    mine.recv_capacity = recv_capacity;
    bool ok_local = narrays >= 1 && narrays <= PCR_HIP_MAX_XFER_ARRAYS && d_send && d_recv && elem_bytes && h_send_counts &&
                    c->world <= PCR_HIP_MAX_ROUTE_PARTS && root >= -1 && root < c->world;
    uint64_t send_total = 0;
    for (int p = 0; ok_local && p < c->world; ++p) { mine.send_counts[p] = h_send_counts[p]; send_total += h_send_counts[p]; }
    for (int a = 0; ok_local && a < narrays; ++a) {
        mine.elem_bytes[a] = elem_bytes[a];
        ok_local = elem_bytes[a] > 0 && elem_bytes[a] <= 16 && (send_total == 0 || d_send[a]) && (recv_capacity == 0 || d_recv[a]);
    }
    mine.valid = ok_local ? 1 : 0;
    std::vector<pcr_hip_xfer_geom> all((size_t)c->world);
    Rccl* r = nullptr;
    int prev = -1;
    PCR_HIP_TRY(hipGetDevice(&prev));
    if (prev != c->device) PCR_HIP_TRY(hipSetDevice(c->device));
    struct Restore { int prev, dev; ~Restore() { if (prev != dev) (void)hipSetDevice(prev); } } restore{prev, c->device};
    if (c->world == 1) all[0] = mine;
    else {
        r = rccl();
        int rc = gather_records(c, r, &mine, all.data(), (size_t)kXferInts, st);
        if (rc) return rc;
    }
    std::vector<uint64_t> send_off((size_t)c->world + 1), recv_cnt((size_t)c->world), recv_off((size_t)c->world + 1);
    int rc = pcr_hip_comm_xfer_plan(all.data(), c->world, c->rank, send_off.data(), recv_cnt.data(), recv_off.data());
    if (rc) return rc;                                            // the same verdict on every rank
    if (h_recv_counts) for (int p = 0; p < c->world; ++p) h_recv_counts[p] = recv_cnt[(size_t)p];
    // my own group never leaves the device
    for (int a = 0; a < narrays; ++a) {
        const size_t eb = (size_t)elem_bytes[a];
        if (recv_cnt[(size_t)c->rank] > 0)
            PCR_HIP_TRY(hipMemcpyAsync(static_cast<char*>(d_recv[a]) + recv_off[(size_t)c->rank] * eb,
                                       static_cast<const char*>(d_send[a]) + send_off[(size_t)c->rank] * eb,
                                       recv_cnt[(size_t)c->rank] * eb, hipMemcpyDeviceToDevice, st));
    }
    if (c->world == 1) return PCR_HIP_OK;
    ncclResult_t first = ncclSuccess;
    const char* where = "";
    auto note = [&](ncclResult_t res, const char* w) { if (res != ncclSuccess && first == ncclSuccess) { first = res; where = w; } };
    PCR_RCCL_TRY(r, r->GroupStart(), "ncclGroupStart");
    for (int a = 0; a < narrays; ++a) {
        const size_t eb = (size_t)elem_bytes[a];
        for (int p = 0; p < c->world; ++p) {
            if (p == c->rank) continue;
            const uint64_t ns = mine.send_counts[p], nr = recv_cnt[(size_t)p];
            if (ns > 0) {
                note(r->Send(static_cast<const char*>(d_send[a]) + send_off[(size_t)p] * eb, ns * eb, ncclInt8, p, c->comm, st), "ncclSend");
                c->bytes_sent += ns * eb;
            }
            if (nr > 0) note(r->Recv(static_cast<char*>(d_recv[a]) + recv_off[(size_t)p] * eb, nr * eb, ncclInt8, p, c->comm, st), "ncclRecv");
        }
    }
    note(r->GroupEnd(), "ncclGroupEnd");
    if (first != ncclSuccess) return rccl_fail(r, first, where);
    return PCR_HIP_OK;
}
}  // namespace

extern "C" {

// The verdict on `world` gathered records + this rank's offsets: a pure function, the same on every rank.
int pcr_hip_comm_xfer_plan(const pcr_hip_xfer_geom* all, int world, int rank, uint64_t* send_offsets, uint64_t* recv_counts,
                           uint64_t* recv_offsets) {
    PCR_REQUIRE(all && world >= 1 && world <= PCR_HIP_MAX_ROUTE_PARTS && rank >= 0 && rank < world,
                "comm_xfer_plan: null records, more than 64 ranks, or rank outside [0, world)");
    PCR_REQUIRE(send_offsets && recv_counts && recv_offsets, "comm_xfer_plan: null result pointer");
    auto who = [](int r) { return "rank " + std::to_string(r); };
    for (int r = 0; r < world; ++r) {
        const pcr_hip_xfer_geom& g = all[r];
        if (!g.valid) return fail(PCR_HIP_INVALID_ARGUMENT, "comm transfer: " + who(r) + " was called with invalid arguments (null arrays, "
                                  "element sizes outside 1..16, more than 8 arrays or 64 ranks); refused on every rank");
        if (g.narrays != all[0].narrays || g.root != all[0].root)
            return fail(PCR_HIP_INVALID_ARGUMENT, "comm transfer: " + who(r) + " brings " + std::to_string(g.narrays) + " arrays for root " +
                        std::to_string(g.root) + " where rank 0 brings " + std::to_string(all[0].narrays) + " for root " +
                        std::to_string(all[0].root) + "; refused on every rank");
        for (int a = 0; a < g.narrays; ++a)
            if (g.elem_bytes[a] != all[0].elem_bytes[a])
                return fail(PCR_HIP_INVALID_ARGUMENT, "comm transfer: array " + std::to_string(a) + " has " + std::to_string(g.elem_bytes[a]) +
                            "-byte elements on " + who(r) + " and " + std::to_string(all[0].elem_bytes[a]) + "-byte elements on rank 0; refused on every rank");
        if (g.root >= 0)
            for (int p = 0; p < world; ++p)
                if (p != g.root && g.send_counts[p] != 0)
                    return fail(PCR_HIP_INVALID_ARGUMENT, "comm_gatherv: " + who(r) + " sends to " + who(p) + ", which is not the root");
    }
    for (int q = 0; q < world; ++q) {
        uint64_t total = 0;
        for (int p = 0; p < world; ++p) total += all[p].send_counts[q];
        if (total > all[q].recv_capacity)
            return fail(PCR_HIP_INVALID_ARGUMENT, "comm transfer: " + who(q) + " would receive " + std::to_string((unsigned long long)total) +
                        " elements per array but has room for " + std::to_string((unsigned long long)all[q].recv_capacity) + "; refused on every rank");
    }
    uint64_t so = 0, ro = 0;
    for (int p = 0; p < world; ++p) {
        send_offsets[p] = so;
        so += all[rank].send_counts[p];
        recv_counts[p] = all[p].send_counts[rank];
        recv_offsets[p] = ro;
        ro += recv_counts[p];
    }
    send_offsets[world] = so;
    recv_offsets[world] = ro;
    return PCR_HIP_OK;
}

int pcr_hip_comm_alltoall_counts(pcr_hip_comm* c, const uint64_t* h_send_counts, uint64_t* h_recv_counts, pcr_hip_stream s) {
    PCR_REQUIRE(c && h_send_counts && h_recv_counts, "comm_alltoall_counts: null argument");
    PCR_REQUIRE(c->world <= PCR_HIP_MAX_ROUTE_PARTS, "comm_alltoall_counts: more than 64 ranks");
    if (c->world == 1) { h_recv_counts[0] = h_send_counts[0]; return PCR_HIP_OK; }
    hipStream_t st = static_cast<hipStream_t>(s);
    int prev = -1;
    PCR_HIP_TRY(hipGetDevice(&prev));
    if (prev != c->device) PCR_HIP_TRY(hipSetDevice(c->device));
    struct Restore { int prev, dev; ~Restore() { if (prev != dev) (void)hipSetDevice(prev); } } restore{prev, c->device};
    const size_t ints = 2 * (size_t)PCR_HIP_MAX_ROUTE_PARTS;
    uint64_t mine[PCR_HIP_MAX_ROUTE_PARTS] = {};
    for (int p = 0; p < c->world; ++p) mine[p] = h_send_counts[p];
    std::vector<uint64_t> all((size_t)c->world * PCR_HIP_MAX_ROUTE_PARTS);
    int rc = gather_records(c, rccl(), mine, all.data(), ints, st);
    if (rc) return rc;
    for (int p = 0; p < c->world; ++p) h_recv_counts[p] = all[(size_t)p * PCR_HIP_MAX_ROUTE_PARTS + (size_t)c->rank];
    return PCR_HIP_OK;
}

int pcr_hip_comm_alltoallv(pcr_hip_comm* c, int narrays, const void* const* d_send, void* const* d_recv, const int32_t* elem_bytes,
                           const uint64_t* h_send_counts, uint64_t recv_capacity, uint64_t* h_recv_counts, pcr_hip_stream s) {
    return xfer(c, narrays, d_send, d_recv, elem_bytes, h_send_counts, recv_capacity, h_recv_counts, -1, s, "comm_alltoallv");
}

int pcr_hip_comm_gatherv(pcr_hip_comm* c, int narrays, const void* const* d_send, void* const* d_recv, const int32_t* elem_bytes,
                         uint64_t send_count, uint64_t recv_capacity, uint64_t* h_recv_counts, int root, pcr_hip_stream s) {
    PCR_REQUIRE(c, "comm_gatherv: null communicator");
    uint64_t counts[PCR_HIP_MAX_ROUTE_PARTS] = {};
    // (a root outside [0, world) is this rank's invalid argument: announced, refused by everyone)
    if (root >= 0 && root < c->world && root < PCR_HIP_MAX_ROUTE_PARTS) counts[root] = send_count;
    return xfer(c, narrays, d_send, d_recv, elem_bytes, counts, c->rank == root ? recv_capacity : 0, h_recv_counts,
                root >= 0 && root < c->world ? root : c->world /* invalid on purpose */, s, "comm_gatherv");
}

}  // extern "C"
