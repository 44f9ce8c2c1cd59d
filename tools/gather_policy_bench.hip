// Micro-benchmark: does the cache policy of a random 16-byte gather change the sustained gather rate on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_policy_bench tools/gather_policy_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int POLICY>
__device__ __forceinline__ u32x4 load16(const void *p) {
    u32x4 v;
    if (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POLICY>
__global__ __launch_bounds__(256) void k_gather(const uint64_t *tab, uint64_t rows, uint64_t iters, uint64_t *out) {
    uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t s = tid * 0x9E3779B97F4A7C15ULL;
    unsigned acc = 0;
    for (uint64_t it = 0; it < iters; ++it) {
        s = mix64(s);
        const uint64_t r = __umul64hi(s, rows);
        u32x4 v = load16<POLICY>(tab + 2 * r);
        acc ^= v.x ^ v.w;
    }
    if (acc == 0x1234567) out[0] = acc;
}
template <int POLICY> double run(const uint64_t *tab, uint64_t rows, uint64_t *out) {
    const int blocks = 256 * 32 / 4; const uint64_t iters = 600;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather<POLICY>), dim3(blocks), dim3(256), 0, 0, tab, rows, iters / 8, out);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k_gather<POLICY>), dim3(blocks), dim3(256), 0, 0, tab, rows, iters, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return (double)blocks * 256 * iters / (ms * 1e-3) / 1e9;
}
int main(int argc, char **argv) {
    double gb = argc > 1 ? atof(argv[1]) : 1.0;
    uint64_t bytes = (uint64_t)(gb * 1e9) & ~255ULL;
    uint64_t *tab, *out;
    if (hipMalloc(&tab, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&out, 64); (void)hipMemset(tab, 0x5a, bytes);
    printf("table %.1f GB, 32 waves/CU, one dependent 16-B gather per lane per iteration (G gathers/s):\n", gb);
    printf("  default %.1f | nt %.1f | sc0 %.1f | sc1 %.1f | sc0 sc1 %.1f | sc0 sc1 nt %.1f | sc1 nt %.1f\n", run<0>(tab, bytes / 16, out), run<1>(tab, bytes / 16, out),
           run<2>(tab, bytes / 16, out), run<3>(tab, bytes / 16, out), run<4>(tab, bytes / 16, out), run<5>(tab, bytes / 16, out), run<6>(tab, bytes / 16, out));
    return 0;
}
