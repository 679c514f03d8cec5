// ubench_lds_atomics2.hip -- follow-up: 64-bit integer, f64 and CAS-loop float adds in LDS on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kThreads = 1024;
constexpr int kCells = 8192;           // 64 KB of 8-byte cells

__device__ __forceinline__ void cas_add_f32(float* p, float v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = *u, assumed;
    do {
        assumed = old;
        old = atomicCAS(u, assumed, __float_as_uint(__uint_as_float(assumed) + v));
    } while (old != assumed);
}

// MODE 0: u64 add, 1: f64 add (unsafeAtomicAdd double), 2: f32 CAS loop, 3: u32 add with return, 4: u64 add + u32 add (fixed-point sum + count)
template <int MODE, int HOT>
__global__ void __launch_bounds__(kThreads) k(int iters, unsigned seed, float* out) {
    __shared__ unsigned long long t[kCells + (MODE == 4 ? kCells / 2 : 0)];
    for (int i = threadIdx.x; i < kCells + (MODE == 4 ? kCells / 2 : 0); i += kThreads) t[i] = 0;
    __syncthreads();
    unsigned s = seed ^ (blockIdx.x * 7919u + threadIdx.x * 104729u);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        unsigned a = HOT ? ((s >> 10) & 255) : ((s >> 10) & (kCells - 1));
        if (MODE == 0) atomicAdd(&t[a], (unsigned long long)s);
        else if (MODE == 1) unsafeAtomicAdd(reinterpret_cast<double*>(&t[a]), 1.0);
        else if (MODE == 2) cas_add_f32(reinterpret_cast<float*>(&t[a]), 1.0f);
        else if (MODE == 3) acc += atomicAdd(reinterpret_cast<unsigned*>(&t[a]), 1u);
        else { atomicAdd(&t[a], (unsigned long long)s); atomicAdd(reinterpret_cast<unsigned*>(&t[kCells]) + a, 1u); }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)t[5] + acc;
}

template <int MODE, int HOT>
void run(const char* name, int iters, float* d_out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE, HOT><<<256, kThreads>>>(10, 1, d_out);
    (void)hipEventRecord(a);
    k<MODE, HOT><<<256, kThreads>>>(iters, 1, d_out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double lane_ops = 256.0 * kThreads * iters;
    printf("%-44s %8.3f ms  %8.1f G updates/s chip  %6.1f cycles/wave-update/CU (2.4GHz)\n", name, ms,
           lane_ops / ms / 1e6, ms * 1e-3 * 2.4e9 / ((double)kThreads / 64 * iters));
}

int main() {
    float* d; (void)hipMalloc(&d, 4096);
    int it = 2000;
    run<0, 0>("u64 add, random 8K cells", it, d);
    run<0, 1>("u64 add, random 256 cells (hot)", it, d);
    run<1, 0>("f64 add, random 8K cells", it, d);
    run<2, 0>("f32 CAS loop, random 8K cells", it, d);
    run<2, 1>("f32 CAS loop, random 256 cells (hot)", it, d);
    run<3, 0>("u32 add RETURN, random 8K cells", it, d);
    run<3, 1>("u32 add RETURN, random 256 cells (hot)", it, d);
    run<4, 0>("u64 add + u32 add (sum+count), random", it, d);
    return 0;
}
