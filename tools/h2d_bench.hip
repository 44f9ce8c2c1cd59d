// H2D bandwidth from page-locked memory: one copy vs the same bytes split over 2 / 4 streams (do several SDMA engines add up?),
// alone and beside a kernel that saturates HBM with random 16-byte gathers.   hipcc --offload-arch=gfx950 -O3 -o h2d_bench h2d_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
__global__ void k_gather(const uint4 *t, uint64_t n, uint64_t iters, uint64_t *sink) {
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL + 1, acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint4 v = t[x % n];
        acc += v.x ^ v.w;
    }
    if (acc == 0x1234567) *sink = acc;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 1340ull << 20;
    void *h = nullptr, *d = nullptr;
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d, bytes));
    std::memset(h, 1, bytes);
    hipStream_t st[4];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const uint64_t tn = (16ull << 30) / 16;
    uint4 *table; uint64_t *sink;
    CK(hipMalloc(&table, tn * 16)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(table, 0, tn * 16));
    hipStream_t ks; CK(hipStreamCreateWithFlags(&ks, hipStreamNonBlocking));
    for (int busy = 0; busy < 2; ++busy)
        for (int parts : {1, 2, 4}) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                if (busy) hipLaunchKernelGGL(k_gather, dim3(256 * 16), dim3(256), 0, ks, table, tn, (uint64_t)1500, sink);
                const double t0 = now();
                const size_t chunk = bytes / parts;
                for (int p = 0; p < parts; ++p) CK(hipMemcpyAsync((char *)d + p * chunk, (char *)h + p * chunk, chunk, hipMemcpyHostToDevice, st[p]));
                for (int p = 0; p < parts; ++p) CK(hipStreamSynchronize(st[p]));
                const double dt = now() - t0;
                if (dt < best) best = dt;
                CK(hipStreamSynchronize(ks));
            }
            std::printf("%s, %d stream(s): %.1f ms -> %.1f GB/s\n", busy ? "beside a gather kernel" : "alone", parts, best * 1e3, bytes / best / 1e9);
        }
    // D2H for completeness
    { const double t0 = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st[0])); CK(hipStreamSynchronize(st[0])); std::printf("D2H alone, 1 stream: %.1f GB/s\n", bytes / (now() - t0) / 1e9); }
    return 0;
}
