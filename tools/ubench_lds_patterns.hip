// ubench_lds_patterns.hip -- what an LDS atomic wave-instruction costs on gfx950 as a function of the ADDRESS PATTERN of its
// 64 lanes (round 4: decides the lane -> cell mapping of the Line tile kernel).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_lds_patterns.hip -o tools/ubench_lds3
//
// OP      0: ds_add_f64   1: ds_add_u32   2: ds_add_f64 + ds_add_u32 (the Line window)   3: ds_add_f32   4: ds_add_u64
// PATTERN 0: 64 random cells                       1: 64 consecutive cells
//         2: 2 runs of 32 consecutive cells        3: 2 runs of 32 cells, row stride `stride` (a steep segment)
//         4: 4 runs of 16 consecutive              5: 8 runs of 8 consecutive
//         6: 2 runs of 32 cells of a shallow segment: x + 1 every lane, + stride every third lane
//         7: 64 cells, row stride `stride` (one column)
// Cells are 8-byte for OP 0, 2 (f64 plane), 4; 4-byte for OP 1, 3; OP 2 keeps the u32 plane behind the f64 plane.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kThreads = 1024;
constexpr int kCells = 4096;

template <int OP, int PATTERN>
__global__ void __launch_bounds__(kThreads) k(int iters, int stride, unsigned seed, float* out, long long* clk) {
    __shared__ double t64[kCells];
    __shared__ unsigned t32[kCells];
    for (int i = threadIdx.x; i < kCells; i += kThreads) { t64[i] = 0.0; t32[i] = 0u; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = seed ^ (blockIdx.x * 7919u + (threadIdx.x >> 6) * 104729u);      // per WAVE stream (run bases are wave-uniform)
    unsigned sl = s ^ (lane * 2654435761u);
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        sl = sl * 1664525u + 1013904223u;
        unsigned a;
        const unsigned r0 = (s >> 8), r1 = (s >> 8) * 2654435761u;
        auto base = [&](int run) { return (unsigned)((r0 + (unsigned)run * (r1 | 1u)) >> 4); };
        if (PATTERN == 0) a = sl >> 10;
        else if (PATTERN == 1) a = base(0) + lane;
        else if (PATTERN == 2) a = base(lane >> 5) + (lane & 31);
        else if (PATTERN == 3) a = base(lane >> 5) + (lane & 31) * stride;
        else if (PATTERN == 4) a = base(lane >> 4) + (lane & 15);
        else if (PATTERN == 5) a = base(lane >> 3) + (lane & 7);
        else if (PATTERN == 6) a = base(lane >> 5) + (lane & 31) + ((lane & 31) / 3) * stride;
        else a = base(0) + lane * stride;
        a &= (kCells - 1);
        if (OP == 0) unsafeAtomicAdd(&t64[a], 1.0);
        else if (OP == 1) atomicAdd(&t32[a], 1u);
        else if (OP == 2) { unsafeAtomicAdd(&t64[a], 1.0); atomicAdd(&t32[a], 1u); }
        else if (OP == 3) unsafeAtomicAdd(reinterpret_cast<float*>(&t32[a]), 1.0f);
        else atomicAdd(reinterpret_cast<unsigned long long*>(&t64[a]), 3ull);
    }
    __syncthreads();
    const long long c1 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = (float)t64[5] + (float)t32[7]; clk[blockIdx.x] = c1 - c0; }
}

template <int OP, int PATTERN>
void run(const char* name, int iters, int stride, float* d_out, long long* d_clk) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<OP, PATTERN><<<256, kThreads>>>(10, stride, 1, d_out, d_clk);
    (void)hipEventRecord(a);
    k<OP, PATTERN><<<256, kThreads>>>(iters, stride, 1, d_out, d_clk);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    long long clk[256];
    (void)hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)clk[i] / 256;
    const double wi = (double)kThreads / 64 * iters * (OP == 2 ? 2 : 1);      // wave-instructions per CU
    // clock64() = s_memtime counts at the shader clock on gfx9; the event time gives the wall clock of the same loop
    printf("%-58s %7.3f ms  %6.2f cyc/wave-instr (s_memtime)  %6.2f at 2.4 GHz wall\n", name, ms, mean / wi, ms * 1e-3 * 2.4e9 / wi);
}

int main(int argc, char** argv) {
    float* d; long long* c;
    (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 256 * 8);
    const int it = 2000;
#define ROW(OP, name)                                                                   \
    run<OP, 0>(name ": 64 random", it, 0, d, c);                                       \
    run<OP, 1>(name ": 64 consecutive", it, 0, d, c);                                  \
    run<OP, 2>(name ": 2 x 32 consecutive", it, 0, d, c);                              \
    run<OP, 4>(name ": 4 x 16 consecutive", it, 0, d, c);                              \
    run<OP, 5>(name ": 8 x 8 consecutive", it, 0, d, c);                               \
    run<OP, 3>(name ": 2 x 32 column, stride 108", it, 108, d, c);                     \
    run<OP, 3>(name ": 2 x 32 column, stride 109", it, 109, d, c);                     \
    run<OP, 3>(name ": 2 x 32 column, stride 113", it, 113, d, c);                     \
    run<OP, 6>(name ": 2 x 32 shallow, stride 108", it, 108, d, c);                    \
    run<OP, 6>(name ": 2 x 32 shallow, stride 109", it, 109, d, c);                    \
    run<OP, 7>(name ": 64 column, stride 109", it, 109, d, c);
    ROW(0, "f64 add")
    ROW(1, "u32 add")
    ROW(2, "f64 + u32")
    ROW(3, "f32 add")
    ROW(4, "u64 add")
    return 0;
}
