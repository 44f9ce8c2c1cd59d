// charon -- MI355X drop-in for `charon dehost` (rmcolq/charon).  C++14 host front end; every per-read
// computation of the hot path runs in libcharon_hip.so behind the C ABI of include/charon_hip.h.
//
// Mirrors (reference paths relative to the reference checkout):
//   * command line: src/main.cpp:49-65, src/dehost_main.cpp:208-312 (same flags, defaults and integer limits;
//     CLI11 itself is not reproduced)
//   * dehost_main / dehost_reads / dehost_paired_reads: src/dehost_main.cpp:314-550
//   * load_index: src/load_index.cpp:8-15, include/index.hpp:122-138 (cereal binary + sdsl::sd_vector, decoded
//     once to plain interleaved words and streamed into HBM)
//   * Result state machine + training: include/result.hpp, include/classify_stats.hpp:34-114,395-584
//   * TSV row: include/read_entry.hpp:322-337
// Host-side columns kept on the CPU as the survey prescribes: mean quality (src/dehost_main.cpp:355-360) and the
// gzip compression ratio (src/utils.cpp:114-124, zlib, computed in an OpenMP loop overlapping nothing yet).
// --extract writes <prefix>_<category>[_1|_2]<ext>.gz like src/dehost_main.cpp:515-536 / include/result.hpp:118-128 (plain gzip
// members; record layout as seqan3's sequence_file_output, which is recalled, not verified).
// `charon classify` (src/classify_main.cpp) shares the loop: call_category for every read, gamma / beta models (--dist), its own defaults.
// .bz2 input: this build's own block-parallel bzip2 decoder (bz2_stream.inc; libbz2 is not in the image).
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <climits>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <deque>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <unordered_set>
#include <utility>
#include <vector>

#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#include <parallel/algorithm>  // __gnu_parallel::sort (libstdc++ parallel mode, OpenMP): the index builder's minimiser sets
#endif

#include "charon_hip.h"
#include "gzip_size.hpp"

#ifndef CHARON_VERSION
#define CHARON_VERSION "charon-mi355x 0.1.0 (reference behaviour: rmcolq/charon @ 2025-07-04)"
#endif

namespace {

// gzip column: bit-exact size emulator (host/gzip_size.hpp) unless CHARON_ZLIB_ONLY is set or the start-up cross-check against
// the linked zlib fails
bool g_gzip_emulator = true;

#include "log_util.inc"
#include "dehost_args.inc"
#include "sdsl_select.inc"
#include "index_file.inc"
#include "inflate_stream.inc"
#include "bz2_stream.inc"
#include "fastx_reader.inc"
#include "result.inc"
#include "dehost.inc"
#include "index_builder.inc"

}  // namespace

std::string g_inflate_stats;
int main(int argc, char **argv) {
    if (argc < 2) { std::cerr << "A subcommand is required\nRun with --help for more information.\n"; return 106; }
    const std::string sub = argv[1];
    if (sub == "-V" || sub == "--version") { std::cout << CHARON_VERSION << std::endl; return 0; }
    if (sub == "-h" || sub == "--help") {
        std::cout << "Charon: Dehost metagenomic reads\nUsage: charon [OPTIONS] SUBCOMMAND\n\nOptions:\n  -h,--help   Print this help message and exit\n  -V,--version   Show version\n\n"
                     "Subcommands:\n  index    Build an index (IBF) for a number of references split into a small number of bins.\n"
                     "  dehost   Dehost read file into host and other using index.\n"
                     "  classify Classify read file using index.\n";
        return 0;
    }
    if (sub == "_gzsize") {  // hidden diagnostic: per record, gzip size by the linked zlib and by the size emulator (no GPU involved)
        try {
            if (argc < 3) return 2;
            if (const char *t = std::getenv("CHARON_READER_THREADS")) g_reader_threads = std::max(1, std::atoi(t));
            BlockReader in(argv[2]);
            RawBlock blk;
            Deflater d;
            std::cout << "selfcheck\t" << (gzip_emulator_self_check() ? 1 : 0) << "\n";
            while (in.next(blk, 1000, 1u << 24))
                for (const RecView &r : blk.recs) {
                    if (r.seq_len == 0) continue;
                    d.load(r, nullptr);
                    const size_t n = r.seq_len;
                    const uint32_t e = d.sizer.size_padded(reinterpret_cast<const uint8_t *>(d.up.data()), n);
                    std::cout << std::string(r.id, r.id_len) << "\t" << d.zlib_size(n) << "\t" << e << "\n";
                }
            return 0;
        } catch (std::exception &e) { std::cerr << "charon: " << e.what() << std::endl; return 1; }
    }
    if (sub == "_bunzip2") {  // hidden diagnostic: decode a .bz2 with this build's decoder to stdout; block statistics on stderr
        try {
            if (argc < 3) return 2;
            if (const char *t = std::getenv("CHARON_READER_THREADS")) g_reader_threads = std::max(1, std::atoi(t));
            const size_t piece = argc > 3 ? (size_t)std::atol(argv[3]) : ((size_t)64 << 20);
            auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            const double t0 = now();
            Bz2Source bz;
            bz.open(argv[2]);
            Slab buf;
            unsigned long long total = 0;
            const bool quiet = std::getenv("CHARON_BUNZIP2_QUIET") != nullptr;
            while (!bz.at_end()) {
                buf.clear();
                bz.fill(buf, piece);
                total += buf.size();
                if (!quiet && !buf.empty() && std::fwrite(buf.data(), 1, buf.size(), stdout) != buf.size()) return 1;
            }
            std::fflush(stdout);
            std::cerr << "bunzip2: " << total << " bytes, " << bz.blocks_counted() << " blocks counted, " << bz.blocks_wasted() << " discarded, " << (now() - t0) << " s\n";
            return 0;
        } catch (std::exception &e) { std::cerr << "charon: " << e.what() << std::endl; return 1; }
    }
    if (sub == "_inflate") {  // hidden diagnostic: inflate a .gz with this build's decoder and with zlib; sizes, checksums, rates
        try {
            if (argc < 3) return 2;
            if (const char *t = std::getenv("CHARON_READER_THREADS")) g_reader_threads = std::max(1, std::atoi(t));
            auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            const size_t piece = argc > 3 ? (size_t)std::atol(argv[3]) : ((size_t)64 << 20);
            unsigned long long total = 0, sum = 1469598103934665603ULL;
            double t0 = now();
            {
                FastInflate fi(argv[2]);
                if (!fi.open()) { std::cerr << "charon: not a mappable gzip file\n"; return 1; }
                Slab buf;
                while (!fi.at_end()) {
                    buf.clear();
                    fi.fill(buf, piece);
                    total += buf.size();
                    sum += buf.empty() ? 0 : (unsigned char)buf[buf.size() / 2];  // (the decoder checks CRC32 / ISIZE itself)
                }
                g_inflate_stats = std::to_string(fi.parallel_rounds()) + " rounds, " + std::to_string(fi.chunks_accepted()) + " chunks counted, " + std::to_string(fi.chunks_discarded()) + " discarded";
            }
            const double t_own = now() - t0;
            std::cout << "parallel: " << g_inflate_stats << "; crc by " << (g_crc_clmul ? "pclmulqdq folding" : "zlib") << "\n";
            unsigned long long ztotal = 0, zsum = 1469598103934665603ULL;
            t0 = now();
            if (!std::getenv("CHARON_SKIP_ZLIB")) {
                gzFile f = gzopen(argv[2], "rb");
                if (!f) return 1;
                gzbuffer(f, 1 << 20);
                std::vector<char> buf(piece + 65536 + 400);
                // the same piece boundaries cannot be reproduced through gzread, so the checksum is over the whole data via one running crc
                unsigned long zc = crc32(0L, Z_NULL, 0);
                for (;;) {
                    const int n = gzread(f, buf.data(), (unsigned)std::min<size_t>(buf.size(), 1u << 30));
                    if (n <= 0) break;
                    ztotal += (unsigned long long)n;
                    zc += (unsigned char)buf[(size_t)n / 2];  // (gzread checks CRC32 / ISIZE itself)
                }
                gzclose(f);
                zsum = zc;
            }
            const double t_z = now() - t0;
            std::cout << "own: " << total << " bytes in " << t_own << " s (" << total / t_own / 1e9 << " GB/s)  zlib: " << ztotal << " bytes in " << t_z << " s ("
                      << ztotal / std::max(t_z, 1e-9) / 1e9 << " GB/s)\n";
            (void)sum; (void)zsum;
            return total == ztotal || std::getenv("CHARON_SKIP_ZLIB") ? 0 : 3;
        } catch (std::exception &e) { std::cerr << "charon: " << e.what() << std::endl; return 1; }
    }
    if (sub == "_gfmt") {  // hidden self-test: format_g6 against printf's %g on random values of every kind a row holds
        const unsigned long long n = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1000000ULL;
        unsigned long long x = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1, bad = 0, fast = 0;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        char a[64], b[64];
        for (unsigned long long i = 0; i < n; ++i) {
            double v;
            const unsigned kind = (unsigned)(rnd() % 8);
            const double u = (double)(rnd() >> 11) / 9007199254740992.0;
            if (kind == 0) v = (double)(float)((double)(rnd() % 5000) / (double)(1 + rnd() % 5000));        // proportions: float(count) / float(n)
            else if (kind == 1) v = (double)((float)(rnd() % 100000) / (float)(1 + rnd() % 100000));
            else if (kind == 2) v = u;                                                                       // probabilities
            else if (kind == 3) v = std::pow(10.0, -12.0 * u) * (double)(rnd() % 10);
            else if (kind == 4) v = (double)(float)(40.0 * u);                                               // mean quality
            else if (kind == 5) v = (double)(rnd() % 2000000) / 2.0 * (rnd() & 1 ? 1.0 : 1e-3);              // exact ties and near-ties
            else if (kind == 6) v = ((double)(rnd() % 1000000) + 0.5) * std::pow(10.0, -(double)(rnd() % 10));
            else { uint64_t bits = rnd(); std::memcpy(&v, &bits, 8); }                                       // anything, NaN and infinities included
            if (rnd() % 16 == 0) v = -v;
            std::snprintf(a, sizeof a, "%g", v);
            const size_t len = format_g6(v, b);
            b[len] = 0;
            if (std::strcmp(a, b) != 0) { if (++bad < 10) std::cout << "mismatch: " << a << " vs " << b << "\n"; }
            const double av = v < 0 ? -v : v;
            if (av >= 1e-4 && av < 999999.0) ++fast;
        }
        std::cout << "format_g6: " << n << " values, " << fast << " in the fast range, " << bad << " mismatches\n";
        return bad ? 1 : 0;
    }
    if (sub == "_efcheck") {  // hidden diagnostic: Elias-Fano arrays written bit by bit in order (add) and by several threads at once (add_at)
        try {
            if (argc < 6) return 2;
            const uint64_t seed = std::strtoull(argv[2], nullptr, 10), nwords = std::strtoull(argv[3], nullptr, 10);
            const double density = std::atof(argv[4]);
            const int threads = std::max(1, std::atoi(argv[5]));
            std::vector<uint64_t> words(nwords);
            uint64_t x = seed * 0x9E3779B97F4A7C15ULL + 1, ones = 0;
            auto rnd = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
            for (uint64_t i = 0; i < nwords; ++i) {
                uint64_t w = 0;
                for (int b = 0; b < 64; ++b) if ((double)(rnd() >> 11) * (1.0 / 9007199254740992.0) < density) w |= 1ULL << b;
                words[i] = w; ones += (uint64_t)__builtin_popcountll(w);
            }
            EliasFanoWriter a, b;
            a.begin(nwords * 64, ones); b.begin(nwords * 64, ones);
            for (uint64_t i = 0; i < nwords; ++i) { uint64_t w = words[i]; while (w) { a.add(i * 64 + (uint64_t)__builtin_ctzll(w)); w &= w - 1; } }
            // the index builder's own routine, in two blocks so that `before` is exercised too
            const uint64_t half = nwords / 2;
            uint64_t got = ef_add_block(b, words.data(), half, 0, 0, threads);
            got += ef_add_block(b, words.data() + half, nwords - half, half * 64, got, threads);
            const bool same = a.low == b.low && a.high == b.high && a.k == ones && got == ones;
            std::cout << "ones " << ones << " wl " << (int)a.wl << " low words " << a.low.size() << " high words " << a.high.size() << " same " << (same ? 1 : 0) << "\n";
            return same ? 0 : 1;
        } catch (std::exception &e) { std::cerr << "charon: " << e.what() << std::endl; return 1; }
    }
    if (sub == "_selmcl") {  // hidden diagnostic: the two select_support_mcl blocks of a bit vector (u64 bit count + words), built or verified (no GPU involved)
        try {
            if (argc < 6) return 2;
            const std::string mode = argv[2];
            const int threads = std::max(1, std::atoi(argv[5]));
            auto slurp = [](const char *path) {
                std::ifstream f(path, std::ios::binary);
                if (!f) throw std::runtime_error(std::string("cannot open ") + path);
                return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            };
            const std::string bv = slurp(argv[3]);
            if (bv.size() < 8) return 2;
            uint64_t nbits = 0;
            std::memcpy(&nbits, bv.data(), 8);
            std::vector<uint64_t> words((nbits + 63) / 64 + 1, 0);
            if (bv.size() - 8 < (nbits + 63) / 64 * 8) return 2;
            std::memcpy(words.data(), bv.data() + 8, (nbits + 63) / 64 * 8);
            if (mode == "build") {
                std::ofstream o(argv[4], std::ios::binary);
                for (int bit = 1; bit >= 0; --bit) { const std::string b = selmcl::build(words.data(), nbits, bit == 1, threads); o.write(b.data(), (std::streamsize)b.size()); }
                return o ? 0 : 1;
            }
            const std::string blocks = slurp(argv[4]);
            selmcl::TailParser tp{blocks.data(), (uint64_t)blocks.size(), 0, 0};
            selmcl::verify(tp, words.data(), nbits, true, threads);
            selmcl::verify(tp, words.data(), nbits, false, threads);
            if (tp.off != tp.len) tp.fail(std::to_string(tp.len - tp.off) + " bytes follow m_high_0_select");
            std::cout << "select blocks agree with the bit vector\n";
            return 0;
        } catch (std::exception &e) { std::cerr << "charon: " << e.what() << std::endl; return 1; }
    }
    if (sub == "_records") {  // hidden diagnostic: dump what the block reader sees (no GPU involved)
        try {
            if (argc < 3) return 2;
            if (const char *t = std::getenv("CHARON_READER_THREADS")) g_reader_threads = std::max(1, std::atoi(t));
            BlockReader in(argv[2]);
            RawBlock blk;
            const size_t max_recs = argc > 3 ? (size_t)std::atol(argv[3]) : 1000, max_bytes = argc > 4 ? (size_t)std::atol(argv[4]) : (1u << 20);
            while (in.next(blk, max_recs, max_bytes))
                for (const RecView &r : blk.recs) {
                    unsigned long long h = 1469598103934665603ULL; long qs = 0;
                    for (uint32_t i = 0; i < r.seq_len; ++i) { h ^= (unsigned char)"ACGTN??"[std::min<unsigned>(g_codes.t[(unsigned char)r.seq[i]], 5)]; h *= 1099511628211ULL; }
                    for (uint32_t i = 0; i < r.qual_len; ++i) qs += r.qual[i] - 33;
                    std::cout << std::string(r.id, r.id_len) << "\t" << r.seq_len << "\t" << r.qual_len << "\t" << qs << "\t" << h << "\n";
                }
            if (std::getenv("CHARON_DIAG_SPLIT")) std::cerr << "split: " << g_split_pieces << " pieces, " << g_split_records << " records taken in parallel\n";
            return 0;
        } catch (std::exception &e) { std::cerr << "charon: " << e.what() << std::endl; return 1; }
    }
    if (sub == "index") {
        IndexArguments iopt;
        try {
            if (const char *e = std::getenv("CHARON_DEVICE")) iopt.device = std::atoi(e);
            if (!parse_index(argc - 2, argv + 2, iopt)) return 0;
        } catch (ParseError &e) {
            std::cerr << e.what() << "\nRun with --help for more information.\n";
            return 105;
        }
        try {
            index_main(iopt);
            return 0;
        } catch (std::exception &e) {
            g_log.error(e.what());
            std::cerr << "charon: " << e.what() << std::endl;
            return 1;
        }
    }
    if (sub != "dehost" && sub != "classify") { std::cerr << "The following argument was not expected: " << sub << "\nRun with --help for more information.\n"; return 109; }
    DehostArguments opt;
    if (sub == "classify") opt.set_classify_defaults();
    try {
        if (const char *e = std::getenv("CHARON_BATCH_READS")) opt.batch_reads = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
        if (const char *e = std::getenv("CHARON_BATCH_BASES")) opt.batch_bases = std::max<uint64_t>(1 << 20, std::strtoull(e, nullptr, 10)) & ~63ULL;
        if (const char *e = std::getenv("CHARON_DEVICE")) opt.device = std::atoi(e);
        if (const char *e = std::getenv("CHARON_MIN_HITS")) opt.min_hits = (uint8_t)std::atoi(e);
        if (!parse_dehost(argc - 2, argv + 2, opt)) return 0;
    } catch (ParseError &e) {
        std::cerr << e.what() << "\nRun with --help for more information.\n";
        return 105;  // CLI11 parse errors exit non-zero through CLI11_PARSE (src/main.cpp:63)
    }
    try {
        dehost_main(opt);  // the reference's subcommand callback discards dehost_main's return value (src/dehost_main.cpp:311)
        return 0;
    } catch (std::exception &e) {
        g_log.error(e.what());
        std::cerr << "charon: " << e.what() << std::endl;
        return 1;
    }
}
