// fake_rccl.cpp -- TEST INFRASTRUCTURE, never shipped: a stand-in for the ten RCCL entry points libpcr_hip.so binds
// (csrc/comm.hip resolves them by dlopen; PCR_HIP_RCCL points it here), so that the library's own multi-rank code --
// pcr_hip_comm_halo_reduce, _alltoallv, _gatherv, the agreements in front of them, pcr::ShardedPipeline on top -- can run
// with SEVERAL RANKS ON ONE GPU.  Real RCCL admits one rank per device, and a GPU box has one device: without this the
// native exchange beyond world 1 is first executed by the driver's 8-GPU run.  What this double does NOT test is RCCL.
//
// Transport: every message is a file in a directory named by the unique id (written under a temporary name, renamed into
// place: a reader never sees half a message), payloads staged through host memory.  Stream semantics: an operation
// synchronizes the caller's stream, then copies synchronously -- coarser than RCCL's stream ordering, never weaker.
// Group semantics: calls between GroupStart and GroupEnd are queued; GroupEnd posts all sends, then completes all
// receives (messages between a pair of ranks are matched in posting order, as RCCL matches them).
// A receive that is not answered within 120 s fails (ncclSystemError) instead of hanging.
//
// Built by tests/test_gpu_native_multirank.py: g++ -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ... -lamdhip64
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Op {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    hipStream_t stream;
};

struct Comm {
    int rank = 0, world = 1;
    std::string dir;
    std::vector<unsigned long> sent, recvd;      // per peer: messages posted / consumed so far
    unsigned long coll = 0;                      // collectives so far (all ranks call them in the same order)
};

thread_local int g_depth = 0;
thread_local std::vector<std::pair<Comm*, Op>> g_queue;

size_t dtype_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

bool write_file(const std::string& path, const void* data, size_t bytes) {
    const std::string tmp = path + ".part";
    std::FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = bytes == 0 || std::fwrite(data, 1, bytes, f) == bytes;
    std::fclose(f);
    return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}

bool read_file_when_there(const std::string& path, void* data, size_t bytes, bool remove_after) {
    const auto t0 = std::chrono::steady_clock::now();
    struct stat st;
    while (stat(path.c_str(), &st) != 0) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if ((size_t)st.st_size != bytes) return false;           // the sizes of a matched pair must agree, as in RCCL
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool ok = bytes == 0 || std::fread(data, 1, bytes, f) == bytes;
    std::fclose(f);
    if (remove_after) std::remove(path.c_str());
    return ok;
}

ncclResult_t do_send(Comm* c, const Op& op) {
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> host(op.bytes);
    if (op.bytes && hipMemcpy(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const std::string path = c->dir + "/m_" + std::to_string(c->rank) + "_" + std::to_string(op.peer) + "_" + std::to_string(c->sent[op.peer]++);
    return write_file(path, host.data(), op.bytes) ? ncclSuccess : ncclSystemError;
}

ncclResult_t do_recv(Comm* c, const Op& op) {
    std::vector<char> host(op.bytes);
    const std::string path = c->dir + "/m_" + std::to_string(op.peer) + "_" + std::to_string(c->rank) + "_" + std::to_string(c->recvd[op.peer]++);
    if (!read_file_when_there(path, host.data(), op.bytes, true)) return ncclSystemError;
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (op.bytes && hipMemcpy(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

// every rank's block to every rank, through files c_<n>_<rank>
ncclResult_t exchange_blocks(Comm* c, const void* d_mine, size_t bytes, std::vector<char>& all, hipStream_t stream) {
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> mine(bytes);
    if (bytes && hipMemcpy(mine.data(), d_mine, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const unsigned long n = c->coll++;
    if (!write_file(c->dir + "/c_" + std::to_string(n) + "_" + std::to_string(c->rank), mine.data(), bytes)) return ncclSystemError;
    all.resize(bytes * c->world);
    for (int r = 0; r < c->world; ++r)
        if (!read_file_when_there(c->dir + "/c_" + std::to_string(n) + "_" + std::to_string(r), all.data() + bytes * r, bytes, false))
            return ncclSystemError;
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "pcrfake_%ld_%ld", (long)getpid(),
                  (long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    const char* base = std::getenv("PCR_FAKE_RCCL_DIR");
    auto* c = new Comm();
    c->rank = rank;
    c->world = nranks;
    c->dir = std::string(base ? base : "/tmp") + "/" + std::string(id.internal, strnlen(id.internal, sizeof id.internal));
    c->sent.assign(nranks, 0);
    c->recvd.assign(nranks, 0);
    mkdir(c->dir.c_str(), 0700);                               // (whoever comes first)
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete reinterpret_cast<Comm*>(comm);                        // (the test removes the directory)
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t first = ncclSuccess;
    for (auto& q : g_queue)
        if (q.second.send) { ncclResult_t r = do_send(q.first, q.second); if (first == ncclSuccess) first = r; }
    for (auto& q : g_queue)
        if (!q.second.send) { ncclResult_t r = do_recv(q.first, q.second); if (first == ncclSuccess) first = r; }
    g_queue.clear();
    return first;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
    Op op{true, const_cast<void*>(buf), count * dtype_bytes(t), peer, stream};
    if (g_depth > 0) { g_queue.push_back({c, op}); return ncclSuccess; }
    return do_send(c, op);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
    Op op{false, buf, count * dtype_bytes(t), peer, stream};
    if (g_depth > 0) { g_queue.push_back({c, op}); return ncclSuccess; }
    return do_recv(c, op);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    std::vector<char> all;
    ncclResult_t r = exchange_blocks(c, send, count * dtype_bytes(t), all, stream);
    if (r != ncclSuccess) return r;
    return hipMemcpy(recv, all.data(), all.size(), hipMemcpyHostToDevice) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = count * dtype_bytes(t);
    std::vector<char> all;
    ncclResult_t r = exchange_blocks(c, send, bytes, all, stream);
    if (r != ncclSuccess) return r;
    std::vector<char> out(all.begin(), all.begin() + bytes);
    auto fold = [&](auto* acc, const auto* v) {
        for (size_t i = 0; i < count; ++i) acc[i] = op == ncclMax ? (v[i] > acc[i] ? v[i] : acc[i]) : acc[i] + v[i];
    };
    for (int k = 1; k < c->world; ++k) {
        const char* blk = all.data() + bytes * k;
        if (t == ncclInt32) fold(reinterpret_cast<int32_t*>(out.data()), reinterpret_cast<const int32_t*>(blk));
        else if (t == ncclUint32) fold(reinterpret_cast<uint32_t*>(out.data()), reinterpret_cast<const uint32_t*>(blk));
        else if (t == ncclFloat64) fold(reinterpret_cast<double*>(out.data()), reinterpret_cast<const double*>(blk));
        else if (t == ncclFloat32) fold(reinterpret_cast<float*>(out.data()), reinterpret_cast<const float*>(blk));
        else return ncclInvalidArgument;
    }
    if (op != ncclMax && op != ncclSum) return ncclInvalidArgument;
    return hipMemcpy(recv, out.data(), bytes, hipMemcpyHostToDevice) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

const char* ncclGetErrorString(ncclResult_t r) {
    return r == ncclSuccess ? "no error" : r == ncclSystemError ? "fake transport: message missing, late or of the wrong size"
         : r == ncclInvalidArgument ? "fake transport: invalid argument" : "fake transport: HIP error";
}

}  // extern "C"
