// How much does a random 16-byte fetch drag out of HBM on gfx950: 64 bytes or the whole 128-byte L2 line?  (FETCH_SIZE alone cannot tell:
// it counts requests at 64 bytes each.)  Per random 128-byte-aligned line two 16-byte loads, both in flight:
//   same half : +0 and +16   -> one request either way: the control, the plain random-fetch rate
//   two halves: +0 and +64   -> one request if a miss fills the line, two if it fills 64 bytes: the LINE rate halves in that case
// build: hipcc --offload-arch=gfx950 -O3 -o tools/fetch_granularity tools/fetch_granularity.hip      usage: fetch_granularity [GB]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
template <int SECOND>  // byte offset of the second load inside the line; 0 = no second load
__global__ __launch_bounds__(256) void k_lines(const unsigned char *tab, uint64_t lines, uint32_t iters, uint64_t *sink) {
    uint64_t s = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL + 1, acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint64_t z = s + 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
        s = z;  // a chain per thread: no two threads walk the same lines
        const unsigned char *p = tab + __umul64hi(z, lines) * 128;
        const ull2 a = __builtin_nontemporal_load(reinterpret_cast<const ull2 *>(p));
        ull2 b = {0, 0};
        if (SECOND) b = __builtin_nontemporal_load(reinterpret_cast<const ull2 *>(p + SECOND));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc ^= a.x ^ a.y ^ b.x ^ b.y;
    }
    if (acc == 0x123456789ULL) *sink = acc;
}
template <int SECOND> static double run(const unsigned char *tab, uint64_t lines, uint64_t *sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint32_t iters = 1200, blocks = 256 * 4;
    hipLaunchKernelGGL((k_lines<SECOND>), dim3(blocks), dim3(256), 0, 0, tab, lines, iters / 8, sink);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_lines<SECOND>), dim3(blocks), dim3(256), 0, 0, tab, lines, iters, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return (double)blocks * 256 * iters / (ms * 1e-3) / 1e9;
}
int main(int argc, char **argv) {
    const double gb = argc > 1 ? std::atof(argv[1]) : 39.0;
    const uint64_t bytes = (uint64_t)(gb * 1e9) & ~(uint64_t)255;
    unsigned char *tab; uint64_t *sink;
    CK(hipMalloc((void **)&tab, bytes)); CK(hipMalloc((void **)&sink, 8));
    CK(hipMemset(tab, 0x5a, bytes));
    CK(hipDeviceSynchronize());
    const uint64_t lines = bytes / 128;
    for (int rep = 0; rep < 2; ++rep)
        std::printf("%.1f GB table, G lines/s: one 16-byte load per line %.1f | +0 and +16 (same 64-byte half) %.1f | +0 and +32 %.1f | +0 and +64 (the other half) %.1f | +0 and +112 %.1f\n", gb,
                    run<0>(tab, lines, sink), run<16>(tab, lines, sink), run<32>(tab, lines, sink), run<64>(tab, lines, sink), run<112>(tab, lines, sink));
    return 0;
}
