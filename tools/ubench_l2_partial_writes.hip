// ubench_l2_partial_writes.hip -- how many partially written 128-byte lines does an XCD's L2 keep until they fill?
// (round 4: the question behind k_b16_scatter's write amplification -- 2.5 GB leave the L2 for 0.8 GB of 16-byte records
// at 14.5 K tiles, 1.3 GB at 3.2 K.)
//
// The kernel is the scatter pass reduced to its stores: N 16-byte records, each appended to one of T "tiles" chosen at
// random; a tile's range is split into eight sub-ranges, one per virtual XCD (blockIdx % 8), and a record takes the next
// slot of its (virtual XCD, tile) sub-range from a returning atomic.  No input stream at all (the record is made from the
// index), so whatever evicts the open lines here is the write stream itself.  Variants:
//   T          number of tiles = open lines per XCD (x 128 B)
//   RUN        records appended per reservation (a lane stores RUN consecutive slots: 16 x RUN contiguous bytes)
// Time per launch from HIP events; bytes leaving the L2 from  rocprofv3 --pmc WRITE_SIZE  (see tools/ubench_l2_partial.sh).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_l2_partial_writes.hip -o tools/ubench_l2pw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kThreads = 512;
constexpr int kPer = 16;                                  // records per thread, as the scatter pass

template <int RUN>
__global__ void __launch_bounds__(kThreads, 4)
k_partial(unsigned long long n, unsigned T, unsigned cap, unsigned* __restrict__ cursor, uint4* __restrict__ rec) {
    const unsigned vx = blockIdx.x & 7u;
    const unsigned long long base = (unsigned long long)blockIdx.x * (kThreads * (kPer / RUN));     // reservations, not records
    unsigned* mine = cursor + (size_t)vx * T;
    unsigned slot[kPer / RUN], tile[kPer / RUN];
#pragma unroll
    for (int k = 0; k < kPer / RUN; ++k) {
        const unsigned long long i = base + (unsigned long long)k * kThreads + threadIdx.x;
        unsigned h = (unsigned)(i * 2654435761ull >> 13) ^ (unsigned)(i >> 7) * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        tile[k] = (unsigned)(((unsigned long long)h * T) >> 32);
        slot[k] = i * RUN < n ? atomicAdd(&mine[tile[k]], (unsigned)RUN) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 0; k < kPer / RUN; ++k) {
        if (slot[k] == 0xFFFFFFFFu || slot[k] + RUN > cap) continue;
        uint4* dst = rec + ((size_t)tile[k] * 8 + vx) * cap + slot[k];
#pragma unroll
        for (int r = 0; r < RUN; ++r) dst[r] = make_uint4(tile[k], slot[k], (unsigned)r, threadIdx.x);
    }
}

template <int RUN>
void run(unsigned long long n, unsigned T, unsigned* d_cursor, uint4* d_rec, size_t rec_slots) {
    const unsigned cap = (unsigned)(rec_slots / ((size_t)T * 8));
    const unsigned long long launches_n = n / RUN;                 // reservations
    const unsigned blocks = (unsigned)((launches_n + kThreads * (kPer / RUN) - 1) / (kThreads * (kPer / RUN)));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemsetAsync(d_cursor, 0, (size_t)T * 8 * 4);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_partial<RUN>, dim3(blocks), dim3(kThreads), 0, 0, n, T, cap, d_cursor, d_rec);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("tiles %6u  run %d  open lines per XCD %7.2f MB  %7.3f ms  (%.0f M records of 16 B, sub-range capacity %u)\n",
           T, RUN, T * 128.0 / 1e6, best, n / 1e6, cap);
}

int main(int argc, char** argv) {
    const unsigned long long n = 50000000ull;
    const size_t rec_slots = (size_t)n * 2;                        // 2x head room over the expected fill of a sub-range
    unsigned* d_cursor; uint4* d_rec;
    (void)hipMalloc(&d_cursor, (size_t)65536 * 8 * 4);
    if (hipMalloc(&d_rec, rec_slots * sizeof(uint4)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const int only = argc > 1 ? atoi(argv[1]) : 0;
    const unsigned tiles[] = {256, 512, 1024, 2048, 3249, 4096, 8192, 14555, 32768};
    for (unsigned T : tiles) {
        if (only && (unsigned)only != T) continue;
        run<1>(n, T, d_cursor, d_rec, rec_slots);
    }
    for (unsigned T : {3249u, 14555u}) {
        if (only && (unsigned)only != T) continue;
        run<2>(n, T, d_cursor, d_rec, rec_slots);
        run<4>(n, T, d_cursor, d_rec, rec_slots);
    }
    return 0;
}
