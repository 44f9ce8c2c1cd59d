// Micro-benchmark: random gathers of 8/16-byte rows from a large HBM table (the IBF probe access pattern).
// Reports G gathers/s for a given table size, bytes per gather and loads in flight per lane.
// build: hipcc --offload-arch=gfx950 -O3 -o gather_bench tools/gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
template <int BYTES, int ILP>
__global__ __launch_bounds__(256) void k_gather(const uint64_t *tab, uint64_t rows, uint64_t iters, uint64_t *out) {
    uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t acc = 0, s = tid * 0x9E3779B97F4A7C15ULL;
    for (uint64_t it = 0; it < iters; ++it) {
        uint64_t r[ILP];
#pragma unroll
        for (int j = 0; j < ILP; ++j) { s = mix64(s); r[j] = __umul64hi(s, rows); }
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            if (BYTES == 8) acc ^= tab[r[j]];
            else if (BYTES == 128) { acc ^= tab[(r[j] & ~15ULL)]; acc ^= tab[(r[j] & ~15ULL) + 8]; }  // both 64-B halves of one 128-B line
            else { ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(tab + 2 * r[j]); acc ^= v.x & v.y; }
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}
template <int BYTES, int ILP> double run(const uint64_t *tab, uint64_t rows, uint64_t *out, int blocks, uint64_t iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather<BYTES, ILP>), dim3(blocks), dim3(256), 0, 0, tab, rows, iters / 8, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_gather<BYTES, ILP>), dim3(blocks), dim3(256), 0, 0, tab, rows, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return (double)blocks * 256 * iters * ILP / (ms * 1e-3) / 1e9;
}
int main(int argc, char **argv) {
    double gb = argc > 1 ? atof(argv[1]) : 1.0;
    uint64_t bytes = (uint64_t)(gb * 1e9) & ~255ULL;
    uint64_t *tab, *out;
    if (hipMalloc(&tab, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 64);
    hipMemset(tab, 0x5a, bytes);
    if (argc > 2) {  // single configuration (for rocprofv3 --pmc): <GB> <bytes 8|16> : 16 waves/CU, ilp 3
        int blocks = 256 * 16 / 4;
        double g = atoi(argv[2]) == 8 ? run<8, 3>(tab, bytes / 8, out, blocks, 1000) : atoi(argv[2]) == 128 ? run<128, 3>(tab, bytes / 8, out, blocks, 1000) : run<16, 3>(tab, bytes / 16, out, blocks, 1000);
        printf("table %.1f GB %dB ilp3: %.1f Ggather/s, gathers in timed launch: %llu\n", gb, atoi(argv[2]), g, (unsigned long long)blocks * 256 * 1000 * 3);
        return 0;
    }
    for (int wpc : {8, 16, 32}) {  // waves per CU
        int blocks = 256 * wpc / 4;
        printf("table %.1f GB waves/CU %2d | 8B ilp1 %.1f ilp3 %.1f ilp8 %.1f | 16B ilp1 %.1f ilp3 %.1f ilp8 %.1f  Ggather/s\n", gb, wpc,
               run<8, 1>(tab, bytes / 8, out, blocks, 2000), run<8, 3>(tab, bytes / 8, out, blocks, 1000), run<8, 8>(tab, bytes / 8, out, blocks, 400),
               run<16, 1>(tab, bytes / 16, out, blocks, 2000), run<16, 3>(tab, bytes / 16, out, blocks, 1000), run<16, 8>(tab, bytes / 16, out, blocks, 400));
    }
    return 0;
}
