// Micro-benchmark: does the ALLOCATION type of the table (default coarse-grained, fine-grained, uncached MTYPE) change the
// sustained rate / fetch granularity of random 16-byte gathers on gfx950?  (The IBF probe pattern; companion of
// gather_policy_bench.hip, which varies the cache-policy bits of the load on a default allocation.)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_alloc_bench tools/gather_alloc_bench.hip
// usage: gather_alloc_bench <GB> [alloc 0|1|2] [policy 0..3]   (two extra arguments: one configuration, for rocprofv3 --pmc)
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
    if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
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
static double run_policy(int p, const uint64_t *tab, uint64_t rows, uint64_t *out) {
    switch (p) { case 0: return run<0>(tab, rows, out); case 1: return run<1>(tab, rows, out); case 2: return run<2>(tab, rows, out); default: return run<3>(tab, rows, out); }
}
int main(int argc, char **argv) {
    double gb = argc > 1 ? atof(argv[1]) : 1.0;
    uint64_t bytes = (uint64_t)(gb * 1e9) & ~255ULL;
    uint64_t *out;
    (void)hipMalloc(&out, 64);
    const char *anames[3] = {"default (coarse)", "fine-grained", "uncached"};
    const unsigned aflags[3] = {hipDeviceMallocDefault, hipDeviceMallocFinegrained, hipDeviceMallocUncached};
    const char *pnames[4] = {"default", "nt", "sc0 sc1", "sc0 sc1 nt"};
    const int a0 = argc > 3 ? atoi(argv[2]) : 0, a1 = argc > 3 ? a0 + 1 : 3;
    for (int a = a0; a < a1; ++a) {
        uint64_t *tab = nullptr;
        hipError_t e = hipExtMallocWithFlags((void **)&tab, bytes, aflags[a]);
        if (e != hipSuccess) { printf("alloc %s: %s\n", anames[a], hipGetErrorString(e)); continue; }
        (void)hipMemset(tab, 0x5a, bytes);
        (void)hipDeviceSynchronize();
        if (argc > 3) {
            const int p = atoi(argv[3]);
            printf("table %.1f GB alloc %s policy %s: %.1f Ggather/s; gathers in the timed launch: %llu\n", gb, anames[a], pnames[p], run_policy(p, tab, bytes / 16, out),
                   (unsigned long long)(256 * 32 / 4) * 256 * 600);
        } else {
            printf("table %.1f GB alloc %-17s |", gb, anames[a]);
            for (int p = 0; p < 4; ++p) printf(" %s %.1f |", pnames[p], run_policy(p, tab, bytes / 16, out));
            printf("  G gathers/s (16 B, 32 waves/CU)\n");
        }
        (void)hipFree(tab);
    }
    return 0;
}
