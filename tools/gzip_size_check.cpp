// Compares gzsize::GzipSizer (charon_amd/csrc/host/gzip_size.hpp) with the linked zlib on synthetic DNA of many shapes, and times both.
//   g++ -O2 -std=c++14 -Icharon_amd/csrc/host -o /tmp/gzip_size_check tools/gzip_size_check.cpp -lz && /tmp/gzip_size_check [n_cases] [seed]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>
#include <zlib.h>
#include "gzip_size.hpp"

struct Z {
    z_stream zs;
    bool init = false;
    std::vector<unsigned char> out;
    size_t size(const std::string &s) {
        if (!init) { std::memset(&zs, 0, sizeof zs); if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("init"); init = true; }
        else deflateReset(&zs);
        const size_t bound = deflateBound(&zs, (uLong)s.size()) + 64;
        if (out.size() < bound) out.resize(bound);
        zs.next_in = (Bytef *)s.data(); zs.avail_in = (uInt)s.size(); zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
        deflate(&zs, Z_FINISH);
        return out.size() - zs.avail_out;
    }
};

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "exhaustive") {
        // every string over {A,C,G,T,N} up to length 8 (488 280 strings) and every ACGT string of length 9..10
        Z z0; gzsize::GzipSizer g0; long bad0 = 0, n0 = 0;
        const char *al = "ACGTN";
        for (int L = 1; L <= 10; ++L) {
            const int base = L <= 8 ? 5 : 4;
            long total = 1; for (int i = 0; i < L; ++i) total *= base;
            std::string s(L, 'A');
            for (long v = 0; v < total; ++v) {
                long x = v; for (int i = 0; i < L; ++i) { s[i] = al[x % base]; x /= base; }
                ++n0;
                if (z0.size(s) != g0.size(reinterpret_cast<const uint8_t *>(s.data()), s.size())) { if (++bad0 <= 10) std::printf("MISMATCH %s\n", s.c_str()); }
            }
        }
        std::printf("exhaustive: %ld strings, %ld mismatches\n", n0, bad0);
        return bad0 ? 1 : 0;
    }
    const long cases = argc > 1 ? std::atol(argv[1]) : 20000;
    std::mt19937_64 rng(argc > 2 ? std::atoll(argv[2]) : 1);
    Z z;
    gzsize::GzipSizer g;
    long bad = 0;
    double tz = 0, tg = 0;
    size_t bytes = 0;
    const char *alpha = "ACGTN";
    for (long c = 0; c < cases; ++c) {
        int kind = (int)(rng() % 14);
        size_t L;
        const bool only5k = std::getenv("ONLY5K") != nullptr;
        if (only5k) kind = 0;
        const char *fixed = std::getenv("FIXEDLEN");
        switch (only5k ? 3 : fixed ? 9 : rng() % 8) {
            case 9: L = (size_t)std::atol(fixed); kind = 0; break;
            case 0: L = 1 + rng() % 40; break;
            case 1: L = 1 + rng() % 400; break;
            case 2: L = 100 + rng() % 300; break;
            case 3: L = 5000; break;
            case 4: L = 1 + rng() % 60000; break;
            case 5: L = 20000 + rng() % 40000; break;
            case 6: L = 60000 + rng() % 400000; break;   // window slides (DNA fast path only)
            case 7: L = 65000 + rng() % 1200; break;     // around the first slide
            default: L = 500 + rng() % 9500; break;
        }
        std::string s(L, 'A');
        if (kind <= 4) { for (auto &ch : s) ch = alpha[rng() % 4]; }                                       // random ACGT
        else if (kind == 5) { for (auto &ch : s) ch = alpha[rng() % 5]; }                                  // with N
        else if (kind == 6) { const size_t p = 1 + rng() % 12; std::string u(p, 'A'); for (auto &ch : u) ch = alpha[rng() % 4]; for (size_t i = 0; i < L; ++i) s[i] = u[i % p]; }  // tandem repeat
        else if (kind == 7) { for (auto &ch : s) ch = alpha[(rng() % 16) ? 0 : rng() % 4]; }               // low complexity
        else if (kind == 8) {                                                                              // random with copied segments
            for (auto &ch : s) ch = alpha[rng() % 4];
            for (int k = 0; k < 20 && L > 200; ++k) { const size_t len = 10 + rng() % 600, a = rng() % L, b = rng() % L; for (size_t i = 0; i < len && a + i < L && b + i < L; ++i) s[b + i] = s[a + i]; }
        } else if (kind == 9) { for (size_t i = 0; i < L; ++i) s[i] = alpha[(i / (1 + rng() % 3)) % 4]; }  // near-periodic
        else if (kind == 10) { for (auto &ch : s) ch = (char)(rng() % 128); }                              // incompressible 7-bit bytes: stored blocks, several blocks
        else if (kind == 11) { for (auto &ch : s) ch = (char)('a' + rng() % 16); }                         // 16 letters: many literals, block splits, generic path
        else if (kind == 12) { const char c0 = alpha[rng() % 5]; for (auto &ch : s) ch = c0; }                // homopolymer
        else { for (auto &ch : s) ch = alpha[rng() % 4]; if (L > 3) s[rng() % L] = 'x'; }                  // DNA with one foreign letter: generic path on DNA-like data
        if (rng() % 7 == 0) for (int k = 0; k < 5; ++k) { const size_t a = rng() % L; for (size_t i = a; i < L && i < a + 30; ++i) s[i] = 'N'; }
        auto t0 = std::chrono::steady_clock::now();
        const size_t want = z.size(s);
        auto t1 = std::chrono::steady_clock::now();
        const size_t got = g.size(reinterpret_cast<const uint8_t *>(s.data()), s.size());
        auto t2 = std::chrono::steady_clock::now();
        tz += std::chrono::duration<double>(t1 - t0).count();
        tg += std::chrono::duration<double>(t2 - t1).count();
        bytes += L;
        if (got == 0 && L > gzsize::GzipSizer::MAX_BYTES) { continue; }  // long non-DNA input: out of the emulator's scope (caller uses zlib)
        if (want != got) {
            if (++bad <= 10) std::printf("MISMATCH case %ld kind %d L %zu: zlib %zu emulator %zu\n", c, kind, L, want, got);
        }
    }
    std::printf("%ld cases, %ld mismatches; zlib %.1f MB/s, emulator %.1f MB/s (%.2fx)  [zlib %s]\n", cases, bad, bytes / tz / 1e6, bytes / tg / 1e6, tz / tg, zlibVersion());
    return bad ? 1 : 0;
}
