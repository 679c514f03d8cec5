// ubench_lds_region.hip -- does the cost of a 64-lane random LDS atomic depend on WHERE in the 160 KB it lands, and on how
// large the region is?  (round 4: the Line window's sum plane sits above 64 KB)
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_lds_region.hip -o tools/ubench_lds4
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kThreads = 1024;

// OP 0: ds_add_u64, 1: ds_add_u32, 2: ds_add_f64, 3: u64 + u32 (same cell index, the Line window)
template <int OP>
__global__ void __launch_bounds__(kThreads) k(int iters, unsigned base, unsigned cells, unsigned base2, float* out, long long* clk) {
    extern __shared__ unsigned char lds[];
    for (unsigned i = threadIdx.x; i < 160 * 1024 / 4; i += kThreads) reinterpret_cast<unsigned*>(lds)[i] = 0u;
    __syncthreads();
    unsigned s = 12345u ^ (blockIdx.x * 7919u + threadIdx.x * 2654435761u);
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned a = (unsigned)(((unsigned long long)(s >> 8) * cells) >> 24);      // uniform in [0, cells)
        if (OP == 0 || OP == 3) atomicAdd(reinterpret_cast<unsigned long long*>(lds + base) + a, 3ull);
        if (OP == 1) atomicAdd(reinterpret_cast<unsigned*>(lds + base) + a, 1u);
        if (OP == 2) unsafeAtomicAdd(reinterpret_cast<double*>(lds + base) + a, 1.0);
        if (OP == 3) atomicAdd(reinterpret_cast<unsigned*>(lds + base2) + a, 1u);
    }
    __syncthreads();
    const long long c1 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = (float)lds[base + 5]; clk[blockIdx.x] = c1 - c0; }
}

template <int OP>
void run(const char* name, unsigned base, unsigned cells, unsigned base2, float* d_out, long long* d_clk) {
    const int iters = 2000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(kThreads), 160 * 1024, 0, 10, base, cells, base2, d_out, d_clk);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(kThreads), 160 * 1024, 0, iters, base, cells, base2, d_out, d_clk);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double wi = (double)kThreads / 64 * iters * (OP == 3 ? 2 : 1);
    printf("%-64s %7.3f ms  %6.2f cycles / wave-instruction at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / wi);
}

int main() {
    float* d; long long* c;
    (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 256 * 8);
    run<0>("u64  4096 cells at 0", 0, 4096, 0, d, c);
    run<0>("u64  4096 cells at 64 KB", 65536, 4096, 0, d, c);
    run<0>("u64  4096 cells at 120 KB", 122880, 4096, 0, d, c);
    run<0>("u64 11664 cells at 16", 16, 11664, 0, d, c);
    run<0>("u64 11664 cells at 65528", 65528, 11664, 0, d, c);
    run<2>("f64 11664 cells at 16", 16, 11664, 0, d, c);
    run<2>("f64 11664 cells at 65528", 65528, 11664, 0, d, c);
    run<1>("u32  4096 cells at 0", 0, 4096, 0, d, c);
    run<1>("u32 11664 cells at 16", 16, 11664, 0, d, c);
    run<1>("u32 11664 cells at 100 KB", 102400, 11664, 0, d, c);
    run<3>("u64 at 65528 + u32 at 16, 11664 cells (the Line window)", 65528, 11664, 16, d, c);
    run<3>("u64 at 16 + u32 at 100 KB, 11664 cells", 16, 11664, 102400, d, c);
    return 0;
}
