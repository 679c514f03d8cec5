// ubench_lds_atomics.hip -- what an LDS atomic costs on gfx950, by type and address pattern.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_lds_atomics.hip -o /tmp/ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kThreads = 1024;
constexpr int kCells = 16384;          // 64 KB of f32

// MODE 0: f32 add, 1: u32 add, 2: f32 add + second plane (two arrays), 3: i32 max, 4: f32 add with return
// PATTERN 0: random (precomputed per-iteration LCG), 1: lane-linear (conflict-free), 2: all lanes same address,
//         3: random within a 32-cell window per wave (high conflict, few distinct)
template <int MODE, int PATTERN>
__global__ void __launch_bounds__(kThreads) k(int iters, unsigned seed, float* out) {
    __shared__ float t[kCells * (MODE == 2 ? 2 : 1)];
    for (int i = threadIdx.x; i < kCells * (MODE == 2 ? 2 : 1); i += kThreads) t[i] = 0.f;
    __syncthreads();
    unsigned s = seed ^ (blockIdx.x * 7919u + threadIdx.x * 104729u);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        unsigned a;
        if (PATTERN == 0) a = (s >> 10) & (kCells - 1);
        else if (PATTERN == 1) a = (threadIdx.x + it * 64) & (kCells - 1);
        else if (PATTERN == 2) a = (it * 17) & (kCells - 1);
        else a = ((threadIdx.x >> 6) * 1024 + ((s >> 10) & 31)) & (kCells - 1);
        if (MODE == 0) unsafeAtomicAdd(&t[a], 1.0f);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&t[a]), 1u);
        else if (MODE == 2) { unsafeAtomicAdd(&t[a], 1.5f); unsafeAtomicAdd(&t[kCells + a], 1.0f); }
        else if (MODE == 3) atomicMax(reinterpret_cast<int*>(&t[a]), (int)s);
        else acc += unsafeAtomicAdd(&t[a], 1.0f);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = t[5] + acc;
}

template <int MODE, int PATTERN>
void run(const char* name, int iters, float* d_out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, PATTERN><<<256, kThreads>>>(10, 1, d_out);
    hipEventRecord(a);
    k<MODE, PATTERN><<<256, kThreads>>>(iters, 1, d_out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double lane_ops = 256.0 * kThreads * iters * (MODE == 2 ? 2 : 1);
    double wave_instr_per_cu = (double)kThreads / 64 * iters * (MODE == 2 ? 2 : 1);
    printf("%-44s %8.3f ms  %8.1f G lane-atomics/s chip  %6.1f cycles/wave-instr/CU (2.4GHz)\n", name, ms,
           lane_ops / ms / 1e6, ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
}

int main() {
    float* d; hipMalloc(&d, 4096);
    int it = 2000;
    run<0, 0>("f32 add, random 16K cells", it, d);
    run<1, 0>("u32 add, random 16K cells", it, d);
    run<3, 0>("i32 max, random 16K cells", it, d);
    run<4, 0>("f32 add RETURN, random", it, d);
    run<2, 0>("f32 add x2 planes, random", it, d);
    run<0, 1>("f32 add, lane-linear (conflict-free)", it, d);
    run<1, 1>("u32 add, lane-linear (conflict-free)", it, d);
    run<0, 2>("f32 add, all lanes one address", it, d);
    run<1, 2>("u32 add, all lanes one address", it, d);
    run<0, 3>("f32 add, random in 32 cells per wave", it, d);
    run<1, 3>("u32 add, random in 32 cells per wave", it, d);
    return 0;
}
