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
// Not implemented: gamma/beta distributions (rejected like an unknown --dist), .bz2 input.
#include <algorithm>
#include <cerrno>
#include <climits>
#include <condition_variable>
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
#include <sys/stat.h>
#include <unistd.h>
#include <unordered_set>
#include <utility>
#include <vector>

#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "charon_hip.h"

#ifndef CHARON_VERSION
#define CHARON_VERSION "charon-mi355x 0.1.0 (reference behaviour: rmcolq/charon @ 2025-07-04)"
#endif

namespace {

// ---------------------------------------------------------------------------------------------------
// logging (plog stand-in: the log text is not part of the parity contract)
// ---------------------------------------------------------------------------------------------------
struct Logger {
    std::ofstream os;
    int level = 0;  // 0 info, 1 debug, 2 verbose
    void open(const std::string &path, int lvl) { level = lvl; if (!path.empty()) os.open(path, std::ios::app); }
    void line(const char *sev, const std::string &msg) {
        if (!os.is_open()) return;
        char buf[32];
        std::time_t t = std::time(nullptr);
        std::strftime(buf, sizeof buf, "%Y-%m-%d %H:%M:%S", std::localtime(&t));
        os << buf << " " << sev << " " << msg << "\n";
        os.flush();
    }
    void info(const std::string &m) { line("INFO ", m); }
    void error(const std::string &m) { line("ERROR", m); }
    void warn(const std::string &m) { line("WARN ", m); }
    void debug(const std::string &m) { if (level >= 1) line("DEBUG", m); }
};
Logger g_log;

bool ends_with(const std::string &s, const std::string &suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}
bool path_exists(const std::string &p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
bool is_file(const std::string &p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

// ---------------------------------------------------------------------------------------------------
// arguments (include/dehost_arguments.hpp:9-43)
// ---------------------------------------------------------------------------------------------------
struct DehostArguments {
    std::string read_file, read_file2, db, category_to_extract, prefix, dist = "kde", log_file = "charon.log";
    bool is_paired = false, run_extract = false;
    uint8_t chunk_size = 100;
    float lo_hi_threshold = 0.15f;
    uint16_t num_reads_to_fit = 5000;
    float min_quality = 15.0f;
    uint32_t min_length = 140;
    float min_compression = 0.0f;
    uint8_t confidence_threshold = 7;
    float confidence_probability_threshold = 0.0f;
    float host_unique_prop_lo_threshold = 0.05f;
    float min_proportion_difference = 0.04f;
    float min_prob_difference = 0.0f;
    uint8_t threads = 1, verbosity = 0;
    // not in the reference: GPU batching knobs (environment only, so the command line stays identical)
    uint64_t batch_reads = 65536, batch_bases = 1ULL << 30;
    int device = 0;
    uint8_t min_hits = 0;  // StatsModel::min_hits_ is uninitialised in the reference; CHARON_MIN_HITS overrides
    std::map<uint8_t, std::vector<std::string>> extract_category_to_file;  // include/dehost_arguments.hpp:20
};

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };

uint64_t parse_uint(const std::string &name, const std::string &v, uint64_t maxv) {
    if (v.empty() || v[0] == '-' || v[0] == '+') throw ParseError("--" + name + ": could not convert '" + v + "' to an unsigned integer");
    errno = 0;
    char *end = nullptr;
    unsigned long long x = std::strtoull(v.c_str(), &end, 10);
    if (errno || *end) throw ParseError("--" + name + ": could not convert '" + v + "'");
    if (x > maxv) throw ParseError("--" + name + ": value " + v + " out of range (max " + std::to_string(maxv) + ")");
    return x;
}
float parse_float(const std::string &name, const std::string &v) {
    errno = 0;
    char *end = nullptr;
    float x = std::strtof(v.c_str(), &end);
    if (v.empty() || errno || *end) throw ParseError("--" + name + ": could not convert '" + v + "' to a number");
    return x;
}

void print_dehost_help() {
    std::cout << "Dehost read file into host and other using index.\n"
                 "Usage: charon dehost [OPTIONS] <fastaq> [<fastaq>]\n\n"
                 "Positionals:\n  <fastaq> FILE REQUIRED      Fasta/q file\n  <fastaq> FILE               Paired Fasta/q file\n\n"
                 "Options:\n"
                 "  -h,--help                   Print this help message and exit\n"
                 "  --db FILE REQUIRED          Prefix for the index.\n"
                 "  -e,--extract STRING         Reads from this category in the index will be extracted to file.\n"
                 "  -p,--prefix FILE            Prefix for the output files.\n"
                 "  --chunk_size INT=100        Read file is read in chunks of this size, to be processed in parallel within a chunk.\n"
                 "  --lo_hi_threshold FLOAT=0.15\n"
                 "  --num_reads_to_fit INT=5000 Number of reads to use to train each distribution in the model.\n"
                 "  -d,--dist STRING            Probability distribution to use for modelling.\n"
                 "  --min_length INT=140        Minimum read length to classify.\n"
                 "  --min_quality INT=15        Minimum read quality to classify.\n"
                 "  --min_compression FLOAT=0   Minimum read gzip compression ratio to classify.\n"
                 "  --confidence INT=7          Minimum difference between the top 2 unique hit counts.\n"
                 "  --host_unique_prop_lo_threshold INT=0.05\n"
                 "  --min_proportion_diff FLOAT=0.04\n"
                 "  --min_probability_diff FLOAT=0\n"
                 "  --log FILE                  File for log\n"
                 "  -t,--threads INT=1          Maximum number of threads to use.\n"
                 "  -v                          Verbosity of logging. Repeat for increased verbosity\n";
}

// returns false if help was printed
bool parse_dehost(int argc, char **argv, DehostArguments &opt) {
    std::vector<std::string> pos;
    bool have_db = false;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        std::string val;
        bool has_val = false;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            size_t eq = a.find('=');
            if (eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); has_val = true; }
        } else if (a.size() > 2 && a[0] == '-' && a[1] != '-' && (a[1] == 't' || a[1] == 'e' || a[1] == 'p' || a[1] == 'd')) {
            val = a.substr(2); a = a.substr(0, 2); has_val = true;  // -t4
            if (!val.empty() && val[0] == '=') val = val.substr(1);
        }
        auto need = [&]() -> std::string {
            if (has_val) return val;
            if (i + 1 >= argc) throw ParseError(a + ": 1 required");
            return argv[++i];
        };
        if (a == "-h" || a == "--help") { print_dehost_help(); return false; }
        else if (a == "--db") { opt.db = need(); have_db = true; }
        else if (a == "-e" || a == "--extract") opt.category_to_extract = need();
        else if (a == "-p" || a == "--prefix") opt.prefix = need();
        else if (a == "--chunk_size") opt.chunk_size = (uint8_t)parse_uint("chunk_size", need(), 255);
        else if (a == "--lo_hi_threshold") opt.lo_hi_threshold = parse_float("lo_hi_threshold", need());
        else if (a == "--num_reads_to_fit") opt.num_reads_to_fit = (uint16_t)parse_uint("num_reads_to_fit", need(), 65535);
        else if (a == "-d" || a == "--dist") opt.dist = need();
        else if (a == "--min_length") opt.min_length = (uint32_t)parse_uint("min_length", need(), 4294967295ULL);
        else if (a == "--min_quality") opt.min_quality = parse_float("min_quality", need());
        else if (a == "--min_compression") opt.min_compression = parse_float("min_compression", need());
        else if (a == "--confidence") opt.confidence_threshold = (uint8_t)parse_uint("confidence", need(), 255);
        else if (a == "--host_unique_prop_lo_threshold") opt.host_unique_prop_lo_threshold = parse_float("host_unique_prop_lo_threshold", need());
        else if (a == "--min_proportion_diff") opt.min_proportion_difference = parse_float("min_proportion_diff", need());
        else if (a == "--min_probability_diff") opt.min_prob_difference = parse_float("min_probability_diff", need());
        else if (a == "--log") opt.log_file = need();
        else if (a == "-t" || a == "--threads") opt.threads = (uint8_t)parse_uint("threads", need(), 255);
        else if (a.size() >= 2 && a[0] == '-' && a.find_first_not_of('v', 1) == std::string::npos) opt.verbosity = (uint8_t)std::min<size_t>(255, opt.verbosity + a.size() - 1);
        else if (!a.empty() && a[0] == '-' && a.size() > 1) throw ParseError("The following argument was not expected: " + a);
        else pos.push_back(a);
    }
    if (pos.empty()) throw ParseError("<fastaq> is required");
    if (pos.size() > 2) throw ParseError("The following argument was not expected: " + pos[2]);
    if (!have_db) throw ParseError("--db is required");
    for (auto &p : pos) if (!is_file(p)) throw ParseError("<fastaq>: File does not exist: " + p);
    if (!path_exists(opt.db)) throw ParseError("--db: Path does not exist: " + opt.db);               // CLI::ExistingPath
    if (!opt.prefix.empty() && path_exists(opt.prefix)) throw ParseError("--prefix: Path already exists: " + opt.prefix);  // NonexistentPath
    opt.read_file = pos[0];
    if (pos.size() > 1) opt.read_file2 = pos[1];
    return true;
}

// ---------------------------------------------------------------------------------------------------
// index file (cereal binary archive) -> metadata + streamed plain rows
// ---------------------------------------------------------------------------------------------------
struct IndexMeta {
    uint8_t window_size = 0, kmer_size = 0;
    double max_fpr = 0;
    uint8_t num_bins = 0;
    std::vector<std::string> categories;
    std::vector<std::pair<std::string, uint8_t>> filepath_to_bin;
    std::map<uint8_t, std::string> bin_to_category;
    uint32_t num_files = 0;
    std::map<uint8_t, uint64_t> records_per_bin, hashes_per_bin;
    uint64_t bins = 0, technical_bins = 0, bin_size = 0, hash_shift = 0, bin_words = 0, hash_funs = 0;

    uint8_t category_index(const std::string &c) const {  // include/input_summary.hpp:39-45
        for (size_t i = 0; i < categories.size(); ++i) if (categories[i] == c) return (uint8_t)i;
        return 255;
    }
    uint8_t host_category_index() const { return std::min(category_index("human"), category_index("host")); }  // :47-55
    std::string category_name(uint8_t index) const {  // :57-62 (guard is '>' in the reference)
        if (index > categories.size()) return "";
        return categories.at(index);
    }
};

class IndexFile {
    std::ifstream is_;
    std::string path_;
    long long file_size_ = 0;
    template <class T> T pod() {
        T v;
        is_.read(reinterpret_cast<char *>(&v), sizeof(T));
        if (!is_) throw std::runtime_error("index file truncated near offset " + std::to_string((long long)is_.tellg()));
        return v;
    }
    std::string str() {
        uint64_t n = pod<uint64_t>();
        if (n > (1u << 20)) throw std::runtime_error("implausible string length in index file near offset " + std::to_string((long long)is_.tellg()));
        std::string s((size_t)n, '\0');
        if (n) is_.read(&s[0], (std::streamsize)n);
        if (!is_) throw std::runtime_error("index file truncated inside a string");
        return s;
    }
    void int_vector(uint8_t &width, uint64_t &bits, std::vector<uint64_t> &words) {
        const long long at = (long long)is_.tellg();
        width = pod<uint8_t>();
        const float growth = pod<float>();
        const uint64_t n_words = pod<uint64_t>();
        bits = pod<uint64_t>();
        if (width < 1 || width > 64 || growth != 1.5f || n_words > (~0ULL >> 6) || n_words * 64 < bits)
            throw std::runtime_error("sdsl int_vector framing check failed at file offset " + std::to_string(at) +
                                     " (width/growth_factor/word count/bit size do not agree)");
        const long long here = (long long)is_.tellg();
        if (here < 0 || n_words > (uint64_t)(file_size_ - here) / 8) throw std::runtime_error("index file truncated inside an int_vector");
        words.assign(n_words + 1, 0);
        is_.read(reinterpret_cast<char *>(words.data()), (std::streamsize)(n_words * 8));
        if (!is_) throw std::runtime_error("index file truncated inside an int_vector");
    }

public:
    IndexMeta meta;
    uint64_t ef_size = 0, ef_ones = 0, high_bits = 0;
    uint8_t ef_wl = 0;
    std::vector<uint64_t> low, high;
    std::vector<uint64_t> bits_per_bin;  // filled by decode_on_device: set bits per technical bin (self-check iv)

    explicit IndexFile(const std::string &path) : is_(path, std::ios::binary), path_(path) {
        if (!is_) throw std::runtime_error("cannot open index file " + path);
        is_.seekg(0, std::ios::end);
        file_size_ = (long long)is_.tellg();
        is_.seekg(0, std::ios::beg);
        IndexMeta &m = meta;
        m.window_size = pod<uint8_t>();
        m.kmer_size = pod<uint8_t>();
        m.max_fpr = pod<double>();
        m.num_bins = pod<uint8_t>();
        uint64_t n = pod<uint64_t>();
        if (n > 255) throw std::runtime_error("implausible category count");
        for (uint64_t i = 0; i < n; ++i) m.categories.push_back(str());
        n = pod<uint64_t>();
        if (n > 65536) throw std::runtime_error("implausible file count");
        for (uint64_t i = 0; i < n; ++i) { std::string p = str(); uint8_t b = pod<uint8_t>(); m.filepath_to_bin.emplace_back(p, b); }
        n = pod<uint64_t>();
        if (n > 256) throw std::runtime_error("implausible bin_to_category size");
        for (uint64_t i = 0; i < n; ++i) { uint8_t b = pod<uint8_t>(); m.bin_to_category[b] = str(); }
        m.num_files = pod<uint32_t>();
        n = pod<uint64_t>();
        if (n > 256) throw std::runtime_error("implausible records_per_bin size");
        for (uint64_t i = 0; i < n; ++i) { uint8_t b = pod<uint8_t>(); m.records_per_bin[b] = pod<uint64_t>(); }
        n = pod<uint64_t>();
        if (n > 256) throw std::runtime_error("implausible hashes_per_bin size");
        for (uint64_t i = 0; i < n; ++i) { uint8_t b = pod<uint8_t>(); m.hashes_per_bin[b] = pod<uint64_t>(); }
        m.bins = pod<uint64_t>(); m.technical_bins = pod<uint64_t>(); m.bin_size = pod<uint64_t>();
        m.hash_shift = pod<uint64_t>(); m.bin_words = pod<uint64_t>(); m.hash_funs = pod<uint64_t>();
        // loader self-checks
        if (m.bins == 0 || m.bins > 255 || m.bins != m.num_bins) throw std::runtime_error("IBF bin count does not match the input summary");
        if (m.technical_bins != ((m.bins + 63) / 64) * 64 || m.bin_words != m.technical_bins / 64)
            throw std::runtime_error("IBF header inconsistent (technical_bins / bin_words)");
        if (m.bin_size == 0 || m.hash_shift != (uint64_t)__builtin_clzll(m.bin_size)) throw std::runtime_error("IBF header inconsistent (hash_shift != countl_zero(bin_size))");
        if (m.hash_funs < 1 || m.hash_funs > 5) throw std::runtime_error("IBF hash function count out of range");
        if (m.kmer_size < 1 || m.kmer_size > 27 || m.window_size < m.kmer_size) throw std::runtime_error("k/w out of range");
        for (uint64_t b = 0; b < m.bins; ++b) {
            auto it = m.bin_to_category.find((uint8_t)b);
            if (it == m.bin_to_category.end() || m.category_index(it->second) == 255) throw std::runtime_error("bin " + std::to_string(b) + " has no known category");
        }
        ef_size = pod<uint64_t>();
        ef_wl = pod<uint8_t>();
        if (ef_size != m.technical_bins * m.bin_size) throw std::runtime_error("sd_vector size != technical_bins * bin_size");
        uint8_t width; uint64_t bits;
        int_vector(width, bits, low);
        if (ef_wl != 0 && width != ef_wl) throw std::runtime_error("sd_vector: m_low width != m_wl");
        ef_ones = ef_wl ? bits / ef_wl : 0;
        if (ef_wl && bits != ef_ones * ef_wl) throw std::runtime_error("sd_vector: m_low size is not a multiple of m_wl");
        int_vector(width, high_bits, high);
        if (width != 1) throw std::runtime_error("sd_vector: m_high is not a bit vector");
        // the two trailing select_support_mcl structures are not needed and not read
    }

    // Decode the Elias-Fano vector into the plain interleaved rows ON THE DEVICE (chn_index_decode_ef): m_high goes over in
    // slices with the matching part of m_low, so neither the host nor the device ever holds a second copy of the plain
    // index, and the per-one work (rank in m_high, low part, bit set) runs at memory speed instead of ~15 ns per one on a
    // host core (minutes for an index of the published size).  Self-checks: number of ones in m_high == elements of
    // m_low; no position beyond m_size or in a technical bin >= num_bins; set bits after decode == ones (a vector whose
    // positions are not strictly increasing would lose bits); per-bin counts for check (iv).
    void decode_on_device(chn_index *index, uint64_t slice_words = 1ULL << 24) {
        const uint64_t n_high = (high_bits + 63) / 64;
        if (high_bits & 63) high[n_high - 1] &= (1ULL << (high_bits & 63)) - 1;  // bits past the end are not part of the vector
        uint64_t ones_before = 0;
        for (uint64_t w0 = 0; w0 < n_high; w0 += slice_words) {
            const uint64_t nw = std::min(slice_words, n_high - w0);
            uint64_t ones = 0;
            for (uint64_t i = 0; i < nw; ++i) ones += (uint64_t)__builtin_popcountll(high[w0 + i]);
            if (ef_wl && ones_before + ones > ef_ones) throw std::runtime_error("sd_vector: m_high holds more ones than m_low has elements");
            if (ones) {
                const uint64_t elem0 = ones_before & ~63ULL;  // 64 elements always end on a word boundary
                const uint64_t lw0 = elem0 * ef_wl / 64, lw1 = ((ones_before + ones) * ef_wl + 63) / 64;
                uint64_t bad = 0;
                if (chn_index_decode_ef(index, ef_size, ef_wl, high.data() + w0, w0 * 64, nw, ones_before, ef_wl ? low.data() + lw0 : nullptr, elem0,
                                        ef_wl ? lw1 - lw0 : 0, &bad) != CHN_OK)
                    throw std::runtime_error(std::string("device decode of the index failed: ") + chn_last_error());
                if (bad) throw std::runtime_error("index has " + std::to_string(bad) + " set bits beyond m_size or in a technical bin >= num_bins");
            }
            ones_before += ones;
        }
        if (!ef_wl) ef_ones = ones_before;  // no low parts: the count is only in m_high
        if (ones_before != ef_ones) throw std::runtime_error("sd_vector: m_high holds fewer ones than m_low has elements");
        bits_per_bin.assign(meta.technical_bins, 0);
        if (chn_index_bin_popcounts(index, bits_per_bin.data()) != CHN_OK) throw std::runtime_error(std::string("bin popcounts failed: ") + chn_last_error());
        uint64_t total = 0;
        for (uint64_t c : bits_per_bin) total += c;
        if (total != ef_ones) throw std::runtime_error("sd_vector: decoded positions are not strictly increasing (" + std::to_string(total) + " distinct bits for " +
                                                       std::to_string(ef_ones) + " ones)");
    }
};

// ---------------------------------------------------------------------------------------------------
// FASTA / FASTQ block reader (seqan3::sequence_file_input<my_traits> semantics, include/utils.hpp:17-19): format by
// extension, optional .gz, dna5 alphabet (IUPAC -> N, lower case accepted, anything else is a parse error).
// A block is a large slab of the (inflated) file holding whole records; records are views into the slab (multi-line
// sequences are compacted in place), so nothing is copied per record and blocks can be parsed on a reader thread while
// the previous block is packed / compressed / classified.
// ---------------------------------------------------------------------------------------------------
struct RecView {
    const char *id = nullptr, *seq = nullptr, *qual = nullptr;
    uint32_t id_len = 0, seq_len = 0, qual_len = 0;
};
struct RawBlock {
    std::vector<char> buf;
    std::vector<RecView> recs;
};

class BlockReader {
    gzFile f_ = nullptr;
    bool fastq_ = false, eof_ = false;
    std::vector<char> carry_;
    std::string path_;

    static const char *find_eol(const char *p, const char *end) { return static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p))); }

    // Parse one record starting at p (non-destructively first).  Returns the position after the record, or nullptr if the
    // data in [p, end) does not hold the whole record yet.
    char *parse_fastq(char *p, char *end, RecView &r) {
        while (p < end && (*p == '\n' || *p == '\r')) ++p;  // blank lines between records
        if (p >= end) return nullptr;
        if (*p != '@') throw std::runtime_error("FASTQ parse error in " + path_ + ": record does not start with '@'");
        const char *e0 = find_eol(p, end);
        if (!e0) return nullptr;
        // pass 1: locate the line structure
        struct Span { char *b; uint32_t n; };
        Span seq_first{nullptr, 0}; bool multi = false;
        uint64_t seq_len = 0;
        char *q = const_cast<char *>(e0) + 1, *plus = nullptr;
        for (;;) {
            if (q >= end) { if (!eof_) return nullptr; throw std::runtime_error("FASTQ parse error in " + path_ + ": unexpected end of file"); }
            const char *e = find_eol(q, end);
            if (!e) { if (!eof_) return nullptr; e = end; }
            uint32_t n = (uint32_t)(e - q);
            if (n && q[n - 1] == '\r') --n;
            if (n && q[0] == '+') { plus = q; q = (e < end) ? const_cast<char *>(e) + 1 : end; break; }
            if (!seq_first.b) seq_first = Span{q, n}; else if (n) multi = true;
            seq_len += n;
            q = (e < end) ? const_cast<char *>(e) + 1 : end;
        }
        (void)plus;
        char *qual_begin = q;
        uint64_t qual_len = 0; bool qmulti = false; bool first_q = true;
        while (qual_len < seq_len) {
            if (q >= end) { if (!eof_) return nullptr; throw std::runtime_error("FASTQ parse error in " + path_ + ": qualities shorter than sequence"); }
            const char *e = find_eol(q, end);
            if (!e) { if (!eof_) return nullptr; e = end; }
            uint32_t n = (uint32_t)(e - q);
            if (n && q[n - 1] == '\r') --n;
            if (!first_q && n) qmulti = true;
            first_q = false;
            qual_len += n;
            q = (e < end) ? const_cast<char *>(e) + 1 : end;
        }
        if (seq_len == 0 && q < end && *q != '@') {  // zero-length read: its (empty) quality line
            const char *e = find_eol(q, end);
            if (e && (e == q || (e == q + 1 && *q == '\r'))) q = const_cast<char *>(e) + 1;
        }
        // pass 2: views (compacting multi-line records in place)
        uint32_t idn = (uint32_t)(e0 - p - 1);
        if (idn && p[idn] == '\r') --idn;
        r.id = p + 1; r.id_len = idn;
        auto compact = [&](char *from, uint64_t want, bool is_multi) -> const char * {
            if (!is_multi) return from;
            char *dst = from, *src = from;
            uint64_t got = 0;
            while (got < want) {
                const char *e = find_eol(src, end);
                if (!e) e = end;
                uint32_t n = (uint32_t)(e - src);
                if (n && src[n - 1] == '\r') --n;
                if (dst != src) std::memmove(dst, src, n);
                dst += n; got += n;
                src = (e < end) ? const_cast<char *>(e) + 1 : end;
            }
            return from;
        };
        r.seq = seq_first.b ? compact(seq_first.b, seq_len, multi) : p; r.seq_len = (uint32_t)seq_len;
        r.qual = compact(qual_begin, qual_len, qmulti); r.qual_len = (uint32_t)qual_len;
        if (seq_len > 0xFFFFFFFFull) throw std::runtime_error("read longer than 2^32 bases");
        return q;
    }
    char *parse_fasta(char *p, char *end, RecView &r) {
        while (p < end && (*p == '\n' || *p == '\r')) ++p;
        if (p >= end) return nullptr;
        if (*p != '>' && *p != ';') throw std::runtime_error("FASTA parse error in " + path_ + ": record does not start with '>'");
        const char *e0 = find_eol(p, end);
        if (!e0) { if (!eof_) return nullptr; e0 = end; }
        // the record ends at the next line that starts with '>' / ';', or at end of file
        char *q = (e0 < end) ? const_cast<char *>(e0) + 1 : end;
        char *rec_end = nullptr;
        for (char *l = q;;) {
            if (l >= end) { if (!eof_) return nullptr; rec_end = end; break; }
            if (*l == '>' || *l == ';') { rec_end = l; break; }
            const char *e = find_eol(l, end);
            if (!e) { if (!eof_) return nullptr; rec_end = end; break; }
            l = const_cast<char *>(e) + 1;
        }
        uint32_t idn = (uint32_t)(e0 - p - 1);
        if (idn && p[idn] == '\r') --idn;
        r.id = p + 1; r.id_len = idn;
        char *dst = q;
        for (char *c = q; c < rec_end; ++c) {  // seqan3 skips blanks and digits inside FASTA sequence lines
            const char ch = *c;
            if (ch == '\n' || ch == '\r' || ch == ' ' || ch == '\t' || (ch >= '0' && ch <= '9')) continue;
            *dst++ = ch;
        }
        r.seq = q; r.seq_len = (uint32_t)(dst - q); r.qual = nullptr; r.qual_len = 0;
        return rec_end;
    }

public:
    explicit BlockReader(const std::string &path) : path_(path) {
        std::string p = path;
        if (ends_with(p, ".gz")) p.resize(p.size() - 3);
        if (ends_with(p, ".bz2")) throw std::runtime_error("bz2 input is not supported by this build: " + path);
        fastq_ = ends_with(p, ".fastq") || ends_with(p, ".fq");
        if (!fastq_ && !(ends_with(p, ".fasta") || ends_with(p, ".fa") || ends_with(p, ".fna") || ends_with(p, ".ffn") ||
                         ends_with(p, ".faa") || ends_with(p, ".frn") || ends_with(p, ".fas")))
            throw std::runtime_error("unknown sequence file extension: " + path);
        f_ = gzopen(path.c_str(), "rb");
        if (!f_) throw std::runtime_error("cannot open " + path);
        gzbuffer(f_, 1 << 20);
    }
    ~BlockReader() { if (f_) gzclose(f_); }
    BlockReader(const BlockReader &) = delete;
    BlockReader &operator=(const BlockReader &) = delete;

    // Fill `blk` with up to max_recs whole records (reading about max_bytes of new data at a time).  Returns false when
    // the file is exhausted and nothing was produced.
    bool next(RawBlock &blk, size_t max_recs, size_t max_bytes) {
        blk.recs.clear();
        blk.buf.swap(carry_);
        carry_.clear();
        size_t parsed_to = 0;
        for (;;) {
            // (re)parsing only ever starts from scratch while no record has been produced, because producing a record may
            // compact it in place
            if (!eof_) {
                const size_t old = blk.buf.size();
                blk.buf.resize(old + max_bytes);
                size_t got = 0;
                while (got < max_bytes) {
                    const int n = gzread(f_, blk.buf.data() + old + got, (unsigned)std::min<size_t>(max_bytes - got, 1u << 30));
                    if (n <= 0) { eof_ = true; break; }
                    got += (size_t)n;
                }
                blk.buf.resize(old + got);
            }
            char *base = blk.buf.data(), *end = base + blk.buf.size(), *p = base;
            while (blk.recs.size() < max_recs) {
                RecView r;
                char *nx = fastq_ ? parse_fastq(p, end, r) : parse_fasta(p, end, r);
                if (!nx) break;
                blk.recs.push_back(r);
                p = nx;
            }
            parsed_to = (size_t)(p - base);
            if (!blk.recs.empty() || eof_) break;
            // not even one whole record yet: read more and parse again
        }
        // whatever was not consumed goes to the next block
        if (parsed_to < blk.buf.size()) {
            bool only_ws = true;
            for (size_t i = parsed_to; i < blk.buf.size() && only_ws; ++i) only_ws = (blk.buf[i] == '\n' || blk.buf[i] == '\r');
            if (!only_ws) carry_.assign(blk.buf.begin() + (long)parsed_to, blk.buf.end());
        }
        if (blk.recs.empty() && eof_ && carry_.empty()) return false;
        if (blk.recs.empty() && eof_) throw std::runtime_error("parse error in " + path_ + ": trailing data is not a whole record");
        return true;
    }
};

inline int dna_code(char c) {  // 0..3 = ACGT, 4 = N (seqan3 dna5: every other IUPAC letter folds to N)
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}
struct CodeTable {  // 0..3 ACGT, 4 N (other IUPAC letters), 255 illegal
    uint8_t t[256];
    CodeTable() {
        std::memset(t, 255, sizeof t);
        for (const char *c = "NRYSWKMBDHVnryswkmbdhv"; *c; ++c) t[(unsigned char)*c] = 4;
        for (const char *c = "ACGTUacgtu"; *c; ++c) t[(unsigned char)*c] = (uint8_t)dna_code(*c);
    }
};
const CodeTable g_codes;

// get_compression_ratio (src/utils.cpp:114-124) of sequence_to_string(seq) (upper-case dna5 letters, :105-112).
// One z_stream per thread, deflateReset between reads (same output as a fresh deflateInit2, without its allocations).
struct Deflater {
    z_stream zs;
    bool init = false;
    std::vector<unsigned char> out;
    std::string up;
    ~Deflater() { if (init) deflateEnd(&zs); }
    float ratio(const RecView &a, const RecView *b) {
        up.clear();
        up.reserve((size_t)a.seq_len + (b ? b->seq_len : 0));
        for (uint32_t i = 0; i < a.seq_len; ++i) up.push_back("ACGTN"[g_codes.t[(unsigned char)a.seq[i]] & 7]);
        if (b) for (uint32_t i = 0; i < b->seq_len; ++i) up.push_back("ACGTN"[g_codes.t[(unsigned char)b->seq[i]] & 7]);
        if (!init) {
            std::memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
            init = true;
        } else {
            deflateReset(&zs);
        }
        const size_t bound = deflateBound(&zs, (uLong)up.size()) + 64;
        if (out.size() < bound) out.resize(bound);
        zs.next_in = (Bytef *)up.data(); zs.avail_in = (uInt)up.size();
        zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
        deflate(&zs, Z_FINISH);
        const size_t compressed = out.size() - zs.avail_out;
        return static_cast<float>(static_cast<double>(compressed) / static_cast<double>(up.size()));
    }
};

// ---------------------------------------------------------------------------------------------------
// per-read entry as the host sees it (ReadEntry, include/read_entry.hpp:16-64, minus the bit rows)
// ---------------------------------------------------------------------------------------------------
struct Entry {
    std::string read_id;
    uint32_t length = 0, num_hashes = 0;
    float mean_quality = 0, compression = 0;
    std::vector<uint32_t> counts, unique;
    std::vector<double> prob;
    uint8_t call = 255, conf = 0;
    uint32_t model_version = 0;  // version of the KDE models the device used for prob/call/conf
    std::string row;             // TSV row formatted (in parallel) for model_version; empty if not formatted yet
    std::string rec_id, rec_seq, rec_qual, rec2_id, rec2_seq, rec2_qual;  // only kept when --extract is given
    uint32_t row_version = 0;
};

// StatsModel training side (include/classify_stats.hpp:34-114,395-584); the probability/call itself runs on the GPU
struct Training {
    struct Data { bool complete = false, pos_complete = false, neg_complete = false; std::vector<float> pos, neg; };
    std::vector<Data> data;
    std::vector<char> model_ready;
    std::vector<std::vector<float>> k_pos, k_neg;  // current KDE datasets per category (reference iteration order)
    bool ready = false;
    uint32_t version = 0;
    uint16_t num_reads_to_fit;
    float lo_hi_threshold;

    Training(const DehostArguments &opt, uint32_t C, const chn_model &def) : num_reads_to_fit(opt.num_reads_to_fit), lo_hi_threshold(opt.lo_hi_threshold) {
        data.resize(C); model_ready.assign(C, 0);
        for (uint32_t c = 0; c < C; ++c) {
            k_pos.emplace_back(def.pos_data[c], def.pos_data[c] + def.pos_n[c]);
            k_neg.emplace_back(def.neg_data[c], def.neg_data[c] + def.neg_n[c]);
        }
    }
    bool check_status(Data &d) {  // :70-78
        if (d.pos.size() >= num_reads_to_fit) d.pos_complete = true;
        if (d.neg.size() >= num_reads_to_fit) d.neg_complete = true;
        if (d.pos_complete && d.neg_complete) d.complete = true;
        return d.complete;
    }
    bool add_pos(Data &d, float v) { if (d.pos.size() < num_reads_to_fit) d.pos.push_back(v); else check_status(d); return d.complete; }       // :80-90
    bool add_neg(Data &d, float v) { if (d.neg.size() < num_reads_to_fit && v > 0) d.neg.push_back(v); else check_status(d); return d.complete; }  // :92-102
    void train(uint32_t i) {  // Model::train -> train_kde (:341-368): fit() copies the data, h unchanged, no sort
        Data &d = data[i];
        if (d.pos_complete) { k_pos[i] = d.pos; ++version; }
        if (d.neg_complete) { k_neg[i] = d.neg; ++version; }
        model_ready[i] = 1;
        d.pos.clear(); d.neg.clear();
    }
    void check_if_ready() { if (ready) return; for (char r : model_ready) if (!r) return; ready = true; }
    void train_model_at(uint32_t i) { train(i); check_if_ready(); }  // :521-533
    void force_ready() { for (uint32_t i = 0; i < data.size(); ++i) if (!model_ready[i]) train(i); ready = true; }  // :463-473
    bool add_read(const std::vector<float> &props) {  // add_read_to_training_data :535-578
        uint8_t pos_i = 255;
        double max_val = 0.0;
        int num_above = 0;
        for (uint8_t i = 0; i < props.size(); ++i) {
            const float val = props[i];
            if (val > lo_hi_threshold) num_above += 1;
            if (val == max_val) pos_i = 255;
            else if (val > max_val) { pos_i = i; max_val = val; }
        }
        const bool to_pos = (pos_i != 255 && num_above == 1);
        const bool to_neg = to_pos || (num_above == 0);
        if (to_pos) { if (add_pos(data[pos_i], props[pos_i]) && !model_ready[pos_i]) train_model_at(pos_i); }
        if (to_neg)
            for (uint8_t i = 0; i < props.size(); ++i)
                if (i != pos_i) { if (add_neg(data[i], props[i]) && !model_ready[i]) train_model_at(i); }
        return ready;
    }
};

struct IndexMeta;
struct Entry;
void format_row(const IndexMeta &meta, const Entry &e, std::string &out);

#define CHN_CHECK(call)                                                                              \
    do {                                                                                             \
        int _rc = (call);                                                                            \
        if (_rc != CHN_OK) throw std::runtime_error(std::string(#call) + " failed: " + chn_last_error()); \
    } while (0)

// Result (include/result.hpp): cache while training, classify, print, count
class Result {
    const IndexMeta &meta_;
    const DehostArguments &opt_;
    chn_stream *stream_;
    chn_model base_model_;
    Training training_;
    std::vector<Entry> cached_;
    size_t cache_capacity_ = 0;
    uint32_t device_model_version_ = 0;
    std::vector<uint64_t> classified_counts_;
    uint64_t unclassified_ = 0;
    std::ostream &out_;
    bool dehost_;  // call_host (single-end) vs call_category (paired; src/dehost_main.cpp:470,475)
    // extract_handles_ (include/result.hpp:44,80-85): one gz FASTA/FASTQ writer per requested category (two when paired)
    struct ExtractFile { gzFile f; bool fastq; };
    std::map<uint8_t, std::vector<ExtractFile>> extract_;

    // seqan3::sequence_file_output record layout [3P-recall]: FASTQ "@id\nSEQ\n+\nQUAL\n"; FASTA ">id\n" + the sequence in
    // lines of 80 letters.  The sequence is the record's dna5 content, i.e. upper case with every non-ACGT letter as N.
    static void write_record(const ExtractFile &x, const std::string &id, const std::string &seq, const std::string &qual) {
        std::string o;
        o.reserve(id.size() + seq.size() * 2 + 16);
        o += x.fastq ? '@' : '>';
        o += id; o += '\n';
        if (x.fastq) {
            for (char c : seq) o += "ACGTN"[g_codes.t[(unsigned char)c] & 7];
            o += "\n+\n"; o += qual; o += '\n';
        } else {
            for (size_t i = 0; i < seq.size(); ++i) {
                o += "ACGTN"[g_codes.t[(unsigned char)seq[i]] & 7];
                if ((i + 1) % 80 == 0 || i + 1 == seq.size()) o += '\n';
            }
            if (seq.empty()) o += '\n';
        }
        if (gzwrite(x.f, o.data(), (unsigned)o.size()) != (int)o.size()) throw std::runtime_error("write to extract file failed");
    }
    void extract(const Entry &e) {  // extract_read / extract_paired_read (include/result.hpp:118-128)
        auto it = extract_.find(e.call);
        if (it == extract_.end()) return;
        write_record(it->second[0], e.rec_id, e.rec_seq, e.rec_qual);
        if (it->second.size() > 1) write_record(it->second[1], e.rec2_id, e.rec2_seq, e.rec2_qual);
    }

    void push_model_to_device() {
        const uint32_t C = (uint32_t)meta_.categories.size();
        std::vector<const float *> pp(C), np(C);
        std::vector<uint32_t> pn(C), nn(C);
        for (uint32_t c = 0; c < C; ++c) {
            pp[c] = training_.k_pos[c].data(); pn[c] = (uint32_t)training_.k_pos[c].size();
            np[c] = training_.k_neg[c].data(); nn[c] = (uint32_t)training_.k_neg[c].size();
        }
        chn_model m = base_model_;
        m.pos_data = pp.data(); m.pos_n = pn.data(); m.neg_data = np.data(); m.neg_n = nn.data();
        CHN_CHECK(chn_model_set(stream_, &m));
        device_model_version_ = training_.version;
    }
    // bring prob/call/conf of `es` up to the current models (device K3 on the cached counts)
    void reclassify(std::vector<Entry *> &es) {
        if (es.empty()) return;
        if (device_model_version_ != training_.version) push_model_to_device();
        const size_t C = meta_.categories.size();
        const size_t chunk = (size_t)std::min<uint64_t>(1 << 16, opt_.batch_reads);  // <= the stream's max_reads
        for (size_t b = 0; b < es.size(); b += chunk) {
            const size_t n = std::min(chunk, es.size() - b);
            std::vector<uint32_t> nh(n), cnt(n * C), unq(n * C), len(n);
            std::vector<float> mq(n), comp(n);
            std::vector<double> prob(n * C);
            std::vector<uint8_t> call(n), conf(n);
            for (size_t i = 0; i < n; ++i) {
                const Entry &e = *es[b + i];
                nh[i] = e.num_hashes; len[i] = e.length; mq[i] = e.mean_quality; comp[i] = e.compression;
                for (size_t c = 0; c < C; ++c) { cnt[i * C + c] = e.counts[c]; unq[i * C + c] = e.unique[c]; }
            }
            CHN_CHECK(chn_classify_counts(stream_, n, nh.data(), cnt.data(), unq.data(), len.data(), mq.data(), comp.data(), prob.data(), call.data(), conf.data()));
            for (size_t i = 0; i < n; ++i) {
                Entry &e = *es[b + i];
                e.prob.assign(prob.begin() + i * C, prob.begin() + (i + 1) * C);
                e.call = call[i]; e.conf = conf[i]; e.model_version = training_.version; e.row.clear();
            }
        }
    }
    void print(Entry &e) {  // print_assignment_result, include/read_entry.hpp:322-337
        if (e.row.empty() || e.row_version != e.model_version) { format_row(meta_, e, e.row); e.row_version = e.model_version; }
        out_.write(e.row.data(), (std::streamsize)e.row.size());
    }
    void classify_read(Entry &e) {  // include/result.hpp:97-116
        if (e.model_version != training_.version) { std::vector<Entry *> one(1, &e); reclassify(one); }
        print(e);
        if (e.call < 255) classified_counts_[e.call] += 1; else unclassified_ += 1;
        if (!extract_.empty()) extract(e);  // add_read / classify_cache: extract right after classify_read (:131-136,186-195)
    }
    void classify_cache() {  // :181-198
        std::vector<Entry *> stale;
        for (Entry &e : cached_) if (e.model_version != training_.version) stale.push_back(&e);
        reclassify(stale);
        for (Entry &e : cached_) classify_read(e);
        cached_.clear();
    }

public:
    Result(const IndexMeta &meta, const DehostArguments &opt, chn_stream *stream, const chn_model &base, std::ostream &out)
        : meta_(meta), opt_(opt), stream_(stream), base_model_(base), training_(opt, (uint32_t)meta.categories.size(), base),
          classified_counts_(meta.categories.size(), 0), out_(out), dehost_(!opt.is_paired) {
        // cached_reads_.reserve() sits inside the `if (opt.run_extract)` loop (include/result.hpp:80-85): capacity 0 otherwise
        if (opt.run_extract) cache_capacity_ = (size_t)opt.num_reads_to_fit * meta.categories.size() * 4;
        for (const auto &kv : opt.extract_category_to_file)
            for (const std::string &path : kv.second) {
                std::string p = path;
                if (ends_with(p, ".gz")) p.resize(p.size() - 3);
                ExtractFile x;
                x.fastq = ends_with(p, ".fastq") || ends_with(p, ".fq");
                x.f = gzopen(path.c_str(), "wb");
                if (!x.f) throw std::runtime_error("cannot create extract file " + path);
                extract_[kv.first].push_back(x);
            }
    }
    ~Result() { for (auto &kv : extract_) for (ExtractFile &x : kv.second) if (x.f) gzclose(x.f); }
    Result(const Result &) = delete;
    Result &operator=(const Result &) = delete;
    uint32_t current_model_version() const { return training_.version; }
    void ensure_device_model() { if (device_model_version_ != training_.version) push_model_to_device(); }

    void add_read(Entry &e) {  // :130-153 (add_paired_read :155-179 differs only in the extract records)
        if (training_.ready) { classify_read(e); return; }
        bool training_complete = false;
        if (cached_.size() < cache_capacity_) {
            cached_.push_back(e);
            std::vector<float> uprops(e.unique.size());
            for (size_t c = 0; c < uprops.size(); ++c) uprops[c] = static_cast<float>(e.unique[c]) / static_cast<float>(e.num_hashes);
            training_complete = training_.add_read(uprops);
        } else {
            training_.force_ready();  // NB: the read that triggers this is dropped, exactly as in the reference
            training_complete = true;
        }
        if (training_complete) classify_cache();
    }
    void complete() { classify_cache(); }  // :200-202
    void print_summary() {                 // :205-213
        g_log.info("Results summary: ");
        for (size_t i = 0; i < classified_counts_.size(); ++i) g_log.info(meta_.categories[i] + " :\t\t" + std::to_string(classified_counts_[i]));
        g_log.info("unclassified :\t" + std::to_string(unclassified_));
    }
};

// ---------------------------------------------------------------------------------------------------
// batching: pack reads into the 2-bit layout of include/charon_hip.h
// ---------------------------------------------------------------------------------------------------
struct HostBatch {
    RawBlock blk1, blk2;               // records of this batch (views into the blocks' slabs)
    std::vector<std::vector<char>> extra;  // further slabs of mate records when one block did not hold enough of them
    std::vector<uint32_t> keep;        // indices of the records that are classified (zero-length reads are skipped)
    std::vector<uint32_t> bases, nmask, len1, len2;
    std::vector<uint64_t> off1, off2;
    std::vector<float> mq, comp;
    bool any_n = false;
    uint64_t n_bases = 0;

    static uint64_t pad64(uint64_t x) { return (x + 63) & ~63ULL; }
    // returns false on an illegal character
    bool put(const RecView &r, uint64_t off, bool &saw_n) {
        uint32_t *bw = bases.data() + (off >> 4);
        uint32_t *nw = nmask.data() + (off >> 5);
        const unsigned char *sq = reinterpret_cast<const unsigned char *>(r.seq);
        bool ok = true;
        for (uint32_t i = 0; i < r.seq_len; i += 16) {
            uint32_t w = 0, nb = 0;
            const uint32_t m = std::min<uint32_t>(16, r.seq_len - i);
            for (uint32_t j = 0; j < m; ++j) {
                const uint8_t c = g_codes.t[sq[i + j]];
                if (c < 4) w |= (uint32_t)c << (2 * j);
                else if (c == 4) nb |= 1u << j;
                else ok = false;
            }
            bw[i >> 4] = w;
            if (nb) { nw[i >> 5] |= nb << (i & 16); saw_n = true; }
        }
        return ok;
    }
    // layout + parallel packing, mean quality and gzip ratio of the records in blk1 (/blk2)
    void pack(bool paired, int threads, bool skip_compression) {
        const size_t nrec = blk1.recs.size();
        keep.clear();
        for (size_t i = 0; i < nrec; ++i) {
            const uint64_t L = (uint64_t)blk1.recs[i].seq_len + (paired ? blk2.recs[i].seq_len : 0);
            if (L == 0) { g_log.warn("Ignoring read " + std::string(blk1.recs[i].id, blk1.recs[i].id_len) + " as has zero length!"); continue; }  // src/dehost_main.cpp:351-354
            if (L > std::numeric_limits<uint32_t>::max()) { g_log.warn("Ignoring read as too long!"); continue; }
            keep.push_back((uint32_t)i);
        }
        const size_t n = keep.size();
        off1.assign(n, 0); len1.assign(n, 0); mq.assign(n, 0); comp.assign(n, 0);
        if (paired) { off2.assign(n, 0); len2.assign(n, 0); }
        uint64_t cur = 0;
        for (size_t i = 0; i < n; ++i) {
            const uint32_t k = keep[i];
            off1[i] = cur; len1[i] = blk1.recs[k].seq_len; cur += pad64(len1[i]);
            if (paired) { off2[i] = cur; len2[i] = blk2.recs[k].seq_len; cur += pad64(len2[i]); }
        }
        n_bases = std::max<uint64_t>(cur, 64);
        bases.assign(n_bases / 16, 0); nmask.assign(n_bases / 32, 0);
        bool saw_n = false, bad = false;
#pragma omp parallel num_threads(threads)
        {
            Deflater defl;
            bool my_n = false, my_bad = false;
#pragma omp for schedule(dynamic, 16)
            for (long i = 0; i < (long)n; ++i) {
                const RecView &a = blk1.recs[keep[i]];
                const RecView *b = paired ? &blk2.recs[keep[i]] : nullptr;
                if (!put(a, off1[i], my_n)) my_bad = true;
                if (b && !put(*b, off2[i], my_n)) my_bad = true;
                // mean quality (src/dehost_main.cpp:355-360 / :441-450): int sum of phred (char - 33) / count, as float
                int sum = 0;
                size_t cnt = a.qual_len;
                for (uint32_t j = 0; j < a.qual_len; ++j) sum += (int)a.qual[j] - 33;
                if (b) { cnt += b->qual_len; for (uint32_t j = 0; j < b->qual_len; ++j) sum += (int)b->qual[j] - 33; }
                mq[i] = cnt ? static_cast<float>(sum) / static_cast<float>(cnt) : 0.0f;
                if (!skip_compression && !my_bad) comp[i] = defl.ratio(a, b);
            }
#pragma omp critical(batch_flags)
            { saw_n = saw_n || my_n; bad = bad || my_bad; }
        }
        any_n = saw_n;
        if (bad) throw std::runtime_error("parse error: illegal character in a sequence (only IUPAC nucleotide letters are accepted)");
    }
};

// bounded hand-over of parsed blocks from the reader thread
struct BlockQueue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::unique_ptr<HostBatch>> q;
    bool done = false;
    std::string error;
    void push(std::unique_ptr<HostBatch> b) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return q.size() < 2; });
        q.push_back(std::move(b));
        cv.notify_all();
    }
    void finish(const std::string &err) { std::lock_guard<std::mutex> lk(m); done = true; error = err; cv.notify_all(); }
    std::unique_ptr<HostBatch> pop() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !q.empty() || done; });
        if (q.empty()) return nullptr;
        std::unique_ptr<HostBatch> b = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return b;
    }
};

std::string first_token(const char *id, uint32_t n) {  // split(id, " ")[0] (src/utils.cpp:9-20)
    const void *sp = std::memchr(id, ' ', n);
    return std::string(id, sp ? (size_t)(static_cast<const char *>(sp) - id) : (size_t)n);
}

// print_assignment_result (include/read_entry.hpp:322-337).  `os << float/double` at precision(6) in the default
// floatfield is printf("%g"); integers as %u.  Formatting rows with snprintf lets batches be formatted in parallel.
void format_row(const IndexMeta &meta, const Entry &e, std::string &out) {
    char buf[128];
    out.clear();
    out += (e.call == 255 ? "U\t" : "C\t");
    out += e.read_id; out += '\t';
    out += meta.category_name(e.call); out += '\t';
    std::snprintf(buf, sizeof buf, "%u\t%u\t%g\t%d\t%g\t", e.length, e.num_hashes, (double)e.mean_quality, (int)e.conf, (double)e.compression);
    out += buf;
    for (size_t i = 0; i < meta.categories.size(); ++i) {
        const float prop = static_cast<float>(e.counts[i]) / static_cast<float>(e.num_hashes);      // get_proportions :140-150
        const float uprop = static_cast<float>(e.unique[i]) / static_cast<float>(e.num_hashes);
        out += meta.categories[i];
        std::snprintf(buf, sizeof buf, ":%u:%g:%g:%g ", e.counts[i], (double)prop, (double)uprop, e.prob[i]);
        out += buf;
    }
    out += '\n';
}

// loader self-check (v), SURVEY 8(c): if a reference FASTA recorded in the index (filepath_to_bin) is still readable, every
// minimiser of it must be found in its bin.  This is the first-contact test for the third-party behaviour this build
// only recalls (seqan3's hash_and_fit fastrange vs the older modulo variant, seeds, alphabet): a miss is reported loudly
// in the log and on stderr, never guessed around.  Only the first 200 kb of at most 8 files are probed.
void self_check_reference_files(const IndexMeta &meta, chn_stream *stream, const DehostArguments &opt) {
    size_t checked = 0;
    for (const auto &fb : meta.filepath_to_bin) {
        if (checked >= 8) break;
        if (!is_file(fb.first)) continue;
        try {
            BlockReader in(fb.first);
            RawBlock blk;
            if (!in.next(blk, 4, 1 << 20)) continue;
            HostBatch hb;
            hb.blk1.recs = blk.recs;
            for (RecView &r : hb.blk1.recs) r.seq_len = std::min<uint32_t>(r.seq_len, 200000);
            hb.pack(false, 1, true);
            const size_t n = hb.keep.size();
            if (n == 0 || n > opt.batch_reads || hb.n_bases > opt.batch_bases) continue;
            chn_batch bt;
            std::memset(&bt, 0, sizeof bt);
            bt.struct_size = sizeof bt; bt.n_reads = n; bt.n_bases = hb.n_bases; bt.bases2 = hb.bases.data();
            bt.nmask = hb.any_n ? hb.nmask.data() : nullptr; bt.seg1_offset = hb.off1.data(); bt.seg1_length = hb.len1.data();
            CHN_CHECK(chn_batch_submit(stream, &bt));
            const size_t C = meta.categories.size();
            std::vector<uint32_t> nh(n), cnt(n * C), unq(n * C);
            chn_result rs;
            std::memset(&rs, 0, sizeof rs);
            rs.struct_size = sizeof rs; rs.num_hashes = nh.data(); rs.counts = cnt.data(); rs.unique_counts = unq.data();
            CHN_CHECK(chn_batch_wait(stream, &rs));
            const uint8_t cat = meta.category_index(meta.bin_to_category.at(fb.second));
            for (size_t i = 0; i < n; ++i) {
                // counts_[cat] is the best bin of the category: it must be at least what the file's own bin holds = all of them
                if (cnt[i * C + cat] != nh[i]) {
                    const std::string msg = "self-check FAILED: only " + std::to_string(cnt[i * C + cat]) + " of " + std::to_string(nh[i]) +
                                            " minimisers of " + fb.first + " are found in the index -- the hash / alphabet conventions of this build do not match the program that wrote the index";
                    g_log.error(msg);
                    std::fprintf(stderr, "charon: %s\n", msg.c_str());
                    return;
                }
            }
            ++checked;
        } catch (std::exception &e) {
            g_log.debug(std::string("self-check skipped for ") + fb.first + ": " + e.what());
        }
    }
    if (checked) g_log.info("self-check: minimisers of " + std::to_string(checked) + " reference file(s) all found in their bins");
}

int dehost_main(DehostArguments &opt) {
    g_log.open(opt.log_file, opt.verbosity);
    if (!ends_with(opt.db, ".idx")) opt.db += ".idx";                 // src/dehost_main.cpp:489-491
    if (!opt.read_file2.empty()) { opt.is_paired = true; opt.min_length = 80; }  // :493-496
    g_log.info(std::string("Running charon dehost\n\nCharon version: ") + CHARON_VERSION);

    IndexFile file(opt.db);
    const IndexMeta &meta = file.meta;
    g_log.info("Loading index from file " + opt.db);
    const uint8_t host_index = meta.host_category_index();
    if (host_index == 255) {
        g_log.error("Index does not contain 'host' or 'human' as a category ");
        throw std::runtime_error("index does not contain 'host' or 'human' as a category");  // assert in the reference (include/index.hpp:76-78)
    }
    g_log.info("Found host at index " + std::to_string(host_index) + " in the index categories");

    opt.run_extract = !opt.category_to_extract.empty();
    if (opt.run_extract && opt.category_to_extract != "all" &&
        std::find(meta.categories.begin(), meta.categories.end(), opt.category_to_extract) == meta.categories.end()) {
        std::string options;
        for (auto &c : meta.categories) options += c + " ";
        g_log.error("Cannot extract " + opt.category_to_extract + ", please chose one of [ all " + options + "]");
        return 1;  // the reference's callback drops this value: exit status stays 0 (src/dehost_main.cpp:311,513-514)
    }
    if (opt.run_extract) {  // src/dehost_main.cpp:515-536
        if (opt.prefix.empty()) opt.prefix = "charon";
        std::vector<std::string> to_extract;
        if (opt.category_to_extract == "all") to_extract = meta.categories; else to_extract.push_back(opt.category_to_extract);
        // get_extension (src/utils.cpp:126-133): extension of the read file, looking through a trailing .gz
        std::string base = opt.read_file;
        const size_t slash = base.find_last_of('/');
        if (slash != std::string::npos) base = base.substr(slash + 1);
        auto ext_of = [](const std::string &f) { const size_t d = f.find_last_of('.'); return (d == std::string::npos || d == 0) ? std::string() : f.substr(d); };
        std::string extension = ext_of(base);
        if (extension == ".gz") extension = ext_of(base.substr(0, base.size() - 3));
        for (const std::string &category : to_extract) {
            const uint8_t ci = meta.category_index(category);
            if (opt.is_paired) {
                opt.extract_category_to_file[ci].push_back(opt.prefix + "_" + category + "_1" + extension + ".gz");
                opt.extract_category_to_file[ci].push_back(opt.prefix + "_" + category + "_2" + extension + ".gz");
            } else {
                opt.extract_category_to_file[ci].push_back(opt.prefix + "_" + category + extension + ".gz");
            }
        }
    }
    if (opt.dist != "gamma" && opt.dist != "beta" && opt.dist != "kde") {
        g_log.error("Supported distributions are [gamma , beta, kde]");
        return 1;
    }
    if (opt.dist != "kde") {
        g_log.error("this build implements dist=kde only");
        std::fprintf(stderr, "charon: only --dist kde is implemented in the MI355X build\n");
        return 1;
    }

    // index -> HBM
    chn_index_desc d;
    std::memset(&d, 0, sizeof d);
    d.struct_size = sizeof d; d.device = opt.device;
    d.kmer_size = meta.kmer_size; d.window_size = meta.window_size; d.hash_funs = (uint8_t)meta.hash_funs;
    d.num_categories = (uint8_t)meta.categories.size(); d.host_index = host_index;
    d.minimiser_seed = 0x8F3F73B5CF1C9ADEULL;
    d.bins = meta.bins; d.technical_bins = meta.technical_bins; d.bin_size = meta.bin_size; d.hash_shift = meta.hash_shift; d.bin_words = meta.bin_words;
    for (uint64_t b = 0; b < meta.bins; ++b) d.bin_to_category[b] = meta.category_index(meta.bin_to_category.at((uint8_t)b));
    chn_index *index = nullptr;
    CHN_CHECK(chn_index_create(&d, &index));
    file.decode_on_device(index);
    file.low.clear(); file.low.shrink_to_fit(); file.high.clear(); file.high.shrink_to_fit();
    g_log.info("Index loaded");
    // loader self-check (iv), SURVEY 8(c): a bin that received n distinct values through h hash functions should have
    // about S * (1 - exp(-h n / S)) set bits.  Reported, never fatal (hashes_per_bin counts are what `charon index` stored).
    for (uint64_t b = 0; b < meta.bins; ++b) {
        auto it = meta.hashes_per_bin.find((uint8_t)b);
        if (it == meta.hashes_per_bin.end() || it->second == 0) continue;
        const double expect = (double)meta.bin_size * (1.0 - std::exp(-(double)meta.hash_funs * (double)it->second / (double)meta.bin_size));
        const double got = (double)file.bits_per_bin[b];
        if (std::fabs(got - expect) > 0.02 * expect + 64)
            g_log.warn("self-check: bin " + std::to_string(b) + " has " + std::to_string((uint64_t)got) + " set bits, expected about " +
                       std::to_string((uint64_t)expect) + " for its " + std::to_string(it->second) + " hashes");
    }

    chn_stream_cfg cfg;
    cfg.struct_size = sizeof cfg; cfg.flags = 0; cfg.max_reads = opt.batch_reads; cfg.max_bases = opt.batch_bases;
    chn_stream *stream = nullptr;
    CHN_CHECK(chn_stream_create(index, &cfg, &stream));
    chn_model model;
    CHN_CHECK(chn_model_default(&model, d.num_categories, host_index, opt.is_paired ? 1 : 0));
    model.min_quality = opt.min_quality; model.min_length = opt.min_length; model.min_compression = opt.min_compression;
    model.confidence_threshold = (int8_t)opt.confidence_threshold;  // narrowing as in StatsModel (include/classify_stats.hpp:404,497)
    model.confidence_probability_threshold = opt.confidence_probability_threshold;
    model.host_unique_prop_lo_threshold = opt.host_unique_prop_lo_threshold;
    model.min_proportion_difference = opt.min_proportion_difference; model.min_prob_difference = opt.min_prob_difference;
    model.min_hits = opt.min_hits;
    CHN_CHECK(chn_model_set(stream, &model));

    self_check_reference_files(meta, stream, opt);

    std::ios::sync_with_stdio(false);
    Result result(meta, opt, stream, model, std::cout);
    g_log.info("Dehosting file " + opt.read_file + (opt.is_paired ? " and " + opt.read_file2 : ""));

    const size_t C = meta.categories.size();
    const bool skip_compression = std::getenv("CHARON_SKIP_COMPRESSION") != nullptr;  // NOT reference behaviour: prints 0
    if (skip_compression) g_log.warn("CHARON_SKIP_COMPRESSION set: the compression column is 0 (differs from the reference)");

    // reader thread: parses whole-record blocks while the previous batch is packed / compressed / classified / printed
    BlockQueue queue;
    std::thread reader([&]() {
        try {
            BlockReader in1(opt.read_file);
            std::unique_ptr<BlockReader> in2;
            if (opt.is_paired) in2.reset(new BlockReader(opt.read_file2));
            const size_t max_bytes = (size_t)std::min<uint64_t>(256ULL << 20, std::max<uint64_t>(1 << 20, opt.batch_bases));
            for (;;) {
                std::unique_ptr<HostBatch> hb(new HostBatch());
                // a batch may hold at most batch_bases padded bases: bound the record count by the byte budget as well
                if (!in1.next(hb->blk1, opt.batch_reads, max_bytes)) break;
                if (opt.is_paired) {
                    // the second file is simply `take`n in step with the first (src/dehost_main.cpp:413-415)
                    if (!in2->next(hb->blk2, hb->blk1.recs.size(), max_bytes)) break;
                    while (hb->blk2.recs.size() < hb->blk1.recs.size()) {
                        // the byte budget cut the mate block short: read further mate blocks until the counts agree
                        RawBlock more;
                        if (!in2->next(more, hb->blk1.recs.size() - hb->blk2.recs.size(), max_bytes)) break;
                        // moving a vector keeps its heap buffer, so the views into `more.buf` stay valid
                        hb->blk2.recs.insert(hb->blk2.recs.end(), more.recs.begin(), more.recs.end());
                        hb->extra.emplace_back(std::move(more.buf));
                    }
                    if (hb->blk2.recs.size() < hb->blk1.recs.size()) hb->blk1.recs.resize(hb->blk2.recs.size());
                }
                queue.push(std::move(hb));
            }
            queue.finish("");
        } catch (std::exception &e) {
            queue.finish(e.what());
        }
    });

    std::vector<uint32_t> nh, cnt, unq;
    std::vector<double> prob;
    std::vector<uint8_t> call, conf, flags;
    std::vector<Entry> entries;
    std::string failure;
    try {
        while (std::unique_ptr<HostBatch> hbp = queue.pop()) {
            HostBatch &hb = *hbp;
            const size_t nrec = hb.blk1.recs.size();
            if (opt.is_paired) {
                for (size_t i = 0; i < nrec; ++i) {  // pair ids must agree after dropping the last character (:423-430)
                    const RecView &a = hb.blk1.recs[i], &b = hb.blk2.recs[i];
                    const uint32_t la = a.id_len ? a.id_len - 1 : 0, lb = b.id_len ? b.id_len - 1 : 0;
                    if (la != lb || std::memcmp(a.id, b.id, la) != 0) {
                        std::cout.flush();
                        std::cout << std::string(a.id, la) << " " << std::string(b.id, lb);
                        std::cout.flush();
                        std::fprintf(stderr, "terminate called after throwing an instance of 'std::runtime_error'\n  what():  Your pairs don't match for read ids.\n");
                        std::abort();
                    }
                }
            }
            // split the block into GPU batches that respect the stream's capacity
            size_t begin = 0;
            while (begin < nrec) {
                uint64_t bases = 0;
                size_t endi = begin;
                while (endi < nrec && endi - begin < opt.batch_reads) {
                    const uint64_t need = HostBatch::pad64(hb.blk1.recs[endi].seq_len) + (opt.is_paired ? HostBatch::pad64(hb.blk2.recs[endi].seq_len) : 0);
                    if (need > opt.batch_bases) throw std::runtime_error("a read is longer than CHARON_BATCH_BASES");
                    if (bases + need > opt.batch_bases) break;
                    bases += need; ++endi;
                }
                HostBatch sub;
                sub.blk1.recs.assign(hb.blk1.recs.begin() + (long)begin, hb.blk1.recs.begin() + (long)endi);
                if (opt.is_paired) sub.blk2.recs.assign(hb.blk2.recs.begin() + (long)begin, hb.blk2.recs.begin() + (long)endi);
                begin = endi;
                sub.pack(opt.is_paired, opt.threads, skip_compression);
                const size_t n = sub.keep.size();
                if (n == 0) continue;
                result.ensure_device_model();
                const uint32_t version = result.current_model_version();
                chn_batch bt;
                std::memset(&bt, 0, sizeof bt);
                bt.struct_size = sizeof bt; bt.on_device = 0; bt.n_reads = n; bt.n_bases = sub.n_bases;
                bt.bases2 = sub.bases.data(); bt.nmask = sub.any_n ? sub.nmask.data() : nullptr;
                bt.seg1_offset = sub.off1.data(); bt.seg1_length = sub.len1.data();
                bt.seg2_offset = opt.is_paired ? sub.off2.data() : nullptr; bt.seg2_length = opt.is_paired ? sub.len2.data() : nullptr;
                bt.mean_quality = sub.mq.data(); bt.compression = sub.comp.data();
                CHN_CHECK(chn_batch_submit(stream, &bt));
                nh.resize(n); cnt.resize(n * C); unq.resize(n * C); prob.resize(n * C); call.resize(n); conf.resize(n); flags.resize(n);
                chn_result rs;
                std::memset(&rs, 0, sizeof rs);
                rs.struct_size = sizeof rs; rs.on_device = 0;
                rs.num_hashes = nh.data(); rs.counts = cnt.data(); rs.unique_counts = unq.data(); rs.probabilities = prob.data();
                rs.call = call.data(); rs.confidence = conf.data(); rs.flags = flags.data();
                CHN_CHECK(chn_batch_wait(stream, &rs));
                // build the entries and format their rows in parallel ...
                entries.assign(n, Entry());
#pragma omp parallel for num_threads(opt.threads) schedule(static)
                for (long i = 0; i < (long)n; ++i) {
                    Entry &e = entries[i];
                    const RecView &a = sub.blk1.recs[sub.keep[i]];
                    e.read_id = first_token(a.id, a.id_len);
                    e.length = sub.len1[i] + (opt.is_paired ? sub.len2[i] : 0u);
                    e.num_hashes = nh[i]; e.mean_quality = sub.mq[i]; e.compression = sub.comp[i];
                    e.counts.assign(cnt.begin() + i * C, cnt.begin() + (i + 1) * C);
                    e.unique.assign(unq.begin() + i * C, unq.begin() + (i + 1) * C);
                    e.prob.assign(prob.begin() + i * C, prob.begin() + (i + 1) * C);
                    e.call = call[i]; e.conf = conf[i]; e.model_version = version; e.row_version = version;
                    format_row(meta, e, e.row);
                    if (opt.run_extract) {
                        e.rec_id.assign(a.id, a.id_len); e.rec_seq.assign(a.seq, a.seq_len); e.rec_qual.assign(a.qual ? a.qual : "", a.qual_len);
                        if (opt.is_paired) {
                            const RecView &b = sub.blk2.recs[sub.keep[i]];
                            e.rec2_id.assign(b.id, b.id_len); e.rec2_seq.assign(b.seq, b.seq_len); e.rec2_qual.assign(b.qual ? b.qual : "", b.qual_len);
                        }
                    }
                }
                // ... then critical(add_read_to_results): serial, in input order (what the reference does at -t 1)
                for (size_t i = 0; i < n; ++i) result.add_read(entries[i]);
            }
        }
        if (!queue.error.empty()) failure = queue.error;
    } catch (std::exception &e) {
        failure = e.what();
        // drain so that the reader thread can finish
        while (queue.pop()) {}
    }
    reader.join();
    if (!failure.empty()) { std::cout.flush(); throw std::runtime_error(failure); }
    result.complete();
    std::cout.flush();
    result.print_summary();
    chn_stream_destroy(stream);
    chn_index_destroy(index);
    return 0;
}

// ===================================================================================================
// charon index  (src/index_main.cpp; include/index_arguments.hpp:9-27; include/store_index.hpp:12-17)
// ===================================================================================================
struct IndexArguments {
    std::string input_file, prefix, tmp_dir, log_file = "charon.log";
    uint8_t window_size = 41, kmer_size = 19;
    uint64_t bits = 4294967293ULL;  // numeric_limits<uint32_t>::max() - 2
    uint8_t num_hash = 3;
    double max_fpr = 0.01;
    uint8_t threads = 1, verbosity = 0;
    bool optimize = false;
    int device = 0;
};

bool parse_index(int argc, char **argv, IndexArguments &opt) {
    std::vector<std::string> pos;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i], val;
        bool has_val = false;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            const size_t eq = a.find('=');
            if (eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); has_val = true; }
        }
        auto need = [&]() -> std::string {
            if (has_val) return val;
            if (i + 1 >= argc) throw ParseError(a + ": 1 required");
            return argv[++i];
        };
        if (a == "-h" || a == "--help") {
            std::cout << "Build an index (IBF) for a number of references split into a small number of bins.\n"
                         "Usage: charon index [OPTIONS] <input>\n\nPositionals:\n  <input> FILE REQUIRED   Tab separated file with columns for filename and category\n\n"
                         "Options:\n  -w INT=41               Window size for (w,k,s)-minimers (must be <=k).\n  -k INT=19               K-mer size for (w,k,s)-minimers.\n"
                         "  -t,--threads INT=1      Maximum number of threads to use.\n  -p,--prefix FILE        Prefix for the output index.\n"
                         "  --temp DIR              Temporary directory for index construction files.\n  --log FILE              File for log\n"
                         "  --optimize              Compress the number of bins for improved classification run time\n  -v                      Verbosity of logging.\n";
            return false;
        }
        else if (a == "-w") opt.window_size = (uint8_t)parse_uint("w", need(), 255);
        else if (a == "-k") opt.kmer_size = (uint8_t)parse_uint("k", need(), 255);
        else if (a == "-t" || a == "--threads") opt.threads = (uint8_t)parse_uint("threads", need(), 255);
        else if (a == "-p" || a == "--prefix") opt.prefix = need();
        else if (a == "--temp") opt.tmp_dir = need();
        else if (a == "--log") opt.log_file = need();
        else if (a == "--optimize") opt.optimize = true;
        else if (a.size() >= 2 && a[0] == '-' && a.find_first_not_of('v', 1) == std::string::npos) opt.verbosity = (uint8_t)std::min<size_t>(255, opt.verbosity + a.size() - 1);
        else if (!a.empty() && a[0] == '-' && a.size() > 1) throw ParseError("The following argument was not expected: " + a);
        else pos.push_back(a);
    }
    if (pos.empty()) throw ParseError("<input> is required");
    if (pos.size() > 1) throw ParseError("The following argument was not expected: " + pos[1]);
    if (!is_file(pos[0])) throw ParseError("<input>: File does not exist: " + pos[0]);
    if (!opt.prefix.empty() && path_exists(opt.prefix)) throw ParseError("--prefix: Path already exists: " + opt.prefix);
    opt.input_file = pos[0];
    return true;
}

std::string make_absolute(const std::string &p) {
    if (!p.empty() && p[0] == '/') return p;
    char buf[4096];
    if (!::getcwd(buf, sizeof buf)) return p;
    return std::string(buf) + "/" + p;
}

// bin_size_in_bits (src/utils.cpp:75-90)
uint64_t bin_size_in_bits(const IndexArguments &opt, uint64_t num_elements) {
    const double numerator = -static_cast<double>(num_elements * opt.num_hash);
    const double denominator = std::log(1 - std::exp(std::log(opt.max_fpr) / opt.num_hash));
    const double result = std::ceil(numerator / denominator);
    if (result > (double)opt.bits) { g_log.warn("Require more bits than available for max_fpr"); return opt.bits; }
    return (uint64_t)result;
}

// sdsl::sd_vector construction from ascending set-bit positions delivered block-wise, and the cereal
// binary archive of Index::serialize (include/index.hpp:122-131).  The two trailing select_support_mcl structures of
// sd_vector are not written: this build's loader does not need them, the reference's loader does (documented gap).
struct EliasFanoWriter {
    uint64_t size = 0, ones = 0, high_bits = 0, k = 0;
    uint8_t wl = 0;
    std::vector<uint64_t> low, high;
    static unsigned hi(uint64_t x) { return x ? 63u - (unsigned)__builtin_clzll(x) : 0u; }
    void begin(uint64_t universe, uint64_t n_ones) {
        size = universe; ones = n_ones; k = 0;
        unsigned logm = hi(ones) + 1;
        const unsigned logn = hi(size) + 1;
        if (logm == logn) --logm;
        wl = (uint8_t)(logn - logm);
        high_bits = ones + (1ULL << logm);
        low.assign((ones * wl + 63) / 64 + 1, 0);
        high.assign((high_bits + 63) / 64, 0);
    }
    void add(uint64_t pos) {
        if (wl) {
            const uint64_t v = pos & ((1ULL << wl) - 1), bit = k * wl, wd = bit >> 6, sh = bit & 63;
            low[wd] |= v << sh;
            if (sh + wl > 64) low[wd + 1] |= v >> (64 - sh);
        }
        const uint64_t hp = (pos >> wl) + k;
        high[hp >> 6] |= 1ULL << (hp & 63);
        ++k;
    }
};

struct BinWriter {
    std::ofstream os;
    explicit BinWriter(const std::string &path) : os(path, std::ios::binary) { if (!os) throw std::runtime_error("cannot create " + path); }
    template <class T> void pod(const T &v) { os.write(reinterpret_cast<const char *>(&v), sizeof(T)); }
    void str(const std::string &s) { pod<uint64_t>(s.size()); os.write(s.data(), (std::streamsize)s.size()); }
    void int_vector(uint8_t width, uint64_t bit_size, const std::vector<uint64_t> &words) {
        const uint64_t n_words = (bit_size + 63) >> 6;
        pod<uint8_t>(width); pod<float>(1.5f); pod<uint64_t>(n_words); pod<uint64_t>(bit_size);
        os.write(reinterpret_cast<const char *>(words.data()), (std::streamsize)(n_words * 8));
    }
};

int index_main(IndexArguments &opt) {
    g_log.open(opt.log_file, opt.verbosity);
    // src/index_main.cpp:274-291
    if (opt.window_size < opt.kmer_size) throw std::logic_error("W must be greater than K");
    if (opt.kmer_size == 0) throw std::logic_error("K must be a positive integer");
    if (opt.kmer_size > 27) throw std::logic_error("K must be at most 27 for the dna5 alphabet (5^k has to fit 64 bits)");
    opt.input_file = make_absolute(opt.input_file);
    if (!opt.prefix.empty()) opt.prefix += ".idx"; else opt.prefix = opt.input_file + ".idx";
    g_log.info(std::string("Running charon index\n\nCharon version: ") + CHARON_VERSION);

    // parse_input_file (:75-116).  Category order = iteration order of an unordered_set<string>, reproduced by using the same
    // container of the same standard library (libstdc++) -- quirk A.9.
    IndexMeta meta;
    meta.window_size = opt.window_size; meta.kmer_size = opt.kmer_size; meta.max_fpr = opt.max_fpr;
    {
        std::ifstream in(opt.input_file);
        if (!in) { g_log.error("Error opening file " + opt.input_file); return 1; }
        std::unordered_set<std::string> categories;
        std::string line;
        uint8_t next_bin = 0;
        while (std::getline(in, line)) {
            if (line.empty()) continue;
            const size_t tab = line.find('\t');
            if (tab == std::string::npos) continue;
            const std::string path = make_absolute(line.substr(0, tab));
            std::string name = line.substr(tab + 1);
            const size_t tab2 = name.find('\t');
            if (tab2 != std::string::npos) name.resize(tab2);
            meta.bin_to_category[next_bin] = name;
            categories.insert(name);
            meta.filepath_to_bin.emplace_back(path, next_bin);
            if (next_bin == 255) { g_log.warn("User has reached the maximum number of files which is 255 - ignoring any additional lines!"); break; }
            next_bin++;
        }
        meta.num_bins = next_bin;
        meta.categories.insert(meta.categories.end(), categories.begin(), categories.end());
    }
    if (meta.num_bins == 0) throw std::runtime_error("no 'path<TAB>category' lines in " + opt.input_file);
    g_log.info("Found " + std::to_string(meta.filepath_to_bin.size()) + " files corresponding to " + std::to_string(meta.categories.size()) + " categories");

    // count_and_store_hashes (:118-160) with the minimisers computed on the GPU.  A throw-away 1-bin index object carries k, w.
    chn_index_desc d0;
    std::memset(&d0, 0, sizeof d0);
    d0.struct_size = sizeof d0; d0.device = opt.device; d0.kmer_size = opt.kmer_size; d0.window_size = opt.window_size; d0.hash_funs = opt.num_hash;
    d0.num_categories = 1; d0.host_index = 255; d0.minimiser_seed = 0x8F3F73B5CF1C9ADEULL;
    d0.bins = 1; d0.technical_bins = 64; d0.bin_size = 64; d0.hash_shift = (uint64_t)__builtin_clzll(64ULL); d0.bin_words = 1;
    chn_index *probe_index = nullptr;
    CHN_CHECK(chn_index_create(&d0, &probe_index));
    const uint64_t max_bases = 1ULL << 28;
    const uint32_t chunk = 4096;
    chn_stream_cfg cfg;
    cfg.struct_size = sizeof cfg; cfg.flags = 0; cfg.max_reads = max_bases / chunk + 65536; cfg.max_bases = max_bases;
    chn_stream *stream = nullptr;
    CHN_CHECK(chn_stream_create(probe_index, &cfg, &stream));

    std::vector<std::vector<uint64_t>> hashes(meta.num_bins);  // distinct minimisers per bin (the reference spills them to <tmp>/<bin>.min)
    for (const auto &fb : meta.filepath_to_bin) {
        const uint8_t bin = fb.second;
        meta.records_per_bin[bin] += 0;
        BlockReader in(fb.first);
        meta.num_files += 1;
        std::vector<uint64_t> &set = hashes[bin];
        uint64_t record_count = 0;
        std::unique_ptr<HostBatch> hb(new HostBatch());
        std::vector<uint64_t> values;
        while (in.next(hb->blk1, 1u << 20, 64u << 20)) {
            record_count += hb->blk1.recs.size();
            // a chromosome-sized record is handled as pieces of 2^26 bases overlapping by w-1 (same union-of-windows argument
            // as for the 4096-base chunks below), so any record length fits the stream
            std::vector<RecView> pieces;
            uint32_t piece = 1u << 26;
            if (const char *e = std::getenv("CHARON_INDEX_PIECE")) piece = std::max<uint32_t>(4096, (uint32_t)std::strtoul(e, nullptr, 10));  // test hook
            for (const RecView &rv : hb->blk1.recs) {
                if (rv.seq_len <= piece + opt.window_size) { pieces.push_back(rv); continue; }
                for (uint64_t st = 0; st + opt.window_size <= rv.seq_len; st += piece) {
                    RecView pv = rv;
                    pv.seq = rv.seq + st;
                    pv.seq_len = (uint32_t)std::min<uint64_t>(rv.seq_len - st, (uint64_t)piece + opt.window_size - 1);
                    pv.qual = nullptr; pv.qual_len = 0;
                    pieces.push_back(pv);
                }
            }
            size_t begin = 0;
            const size_t nrec = pieces.size();
            while (begin < nrec) {  // sub-batches that fit the stream
                uint64_t bases = 0;
                size_t endi = begin;
                // chunks overlap by w-1 bases and every record adds one partial chunk: keep 1/8 of the capacity in reserve
                while (endi < nrec && endi - begin < 60000 && bases + HostBatch::pad64(pieces[endi].seq_len) <= max_bases - max_bases / 8) { bases += HostBatch::pad64(pieces[endi].seq_len); ++endi; }
                if (endi == begin) throw std::runtime_error("internal error: a piece of " + fb.first + " does not fit the stream");
                HostBatch sub;
                sub.blk1.recs.assign(pieces.begin() + (long)begin, pieces.begin() + (long)endi);
                begin = endi;
                sub.pack(false, opt.threads, true);
                if (sub.keep.empty()) continue;
                // cut every record into chunks of 4096 bases overlapping by w-1: a chunk is created only where a whole window starts,
                // so the union of the chunks' minimisers is exactly the record's minimiser set
                std::vector<uint64_t> coff; std::vector<uint32_t> clen;
                for (size_t i = 0; i < sub.keep.size(); ++i) {
                    const uint64_t L = sub.len1[i], o = sub.off1[i];
                    const uint64_t nwin = L >= opt.window_size ? L - opt.window_size + 1 : 1;
                    const uint64_t nch = (nwin + chunk - 1) / chunk;
                    for (uint64_t c = 0; c < nch; ++c) {
                        const uint64_t st = c * chunk;
                        coff.push_back(o + st);
                        clen.push_back((uint32_t)std::min<uint64_t>(L - st, (uint64_t)chunk + opt.window_size - 1));
                    }
                }
                chn_batch bt;
                std::memset(&bt, 0, sizeof bt);
                bt.struct_size = sizeof bt; bt.n_reads = coff.size(); bt.n_bases = sub.n_bases; bt.bases2 = sub.bases.data();
                bt.nmask = sub.any_n ? sub.nmask.data() : nullptr; bt.seg1_offset = coff.data(); bt.seg1_length = clen.data();
                values.resize(sub.n_bases + (uint64_t)opt.window_size * coff.size());
                uint64_t nv = 0;
                CHN_CHECK(chn_minimisers(stream, &bt, values.data(), values.size(), &nv));
                set.insert(set.end(), values.begin(), values.begin() + (long)nv);
                if (set.size() > (1u << 26)) { std::sort(set.begin(), set.end()); set.erase(std::unique(set.begin(), set.end()), set.end()); }
            }
        }
        std::sort(set.begin(), set.end());
        set.erase(std::unique(set.begin(), set.end()), set.end());
        meta.records_per_bin[bin] += record_count;
        meta.hashes_per_bin[bin] += set.size();
        g_log.info("Added file " + fb.first + " with " + std::to_string(record_count) + " records and " + std::to_string(set.size()) + " hashes to bin " + std::to_string(bin));
    }
    chn_stream_destroy(stream);
    chn_index_destroy(probe_index);

    // optimize_layout (:162-236)
    std::map<uint8_t, std::vector<uint8_t>> bucket_to_bins;
    if (!opt.optimize) {
        for (unsigned b = 0; b < meta.num_bins; ++b) bucket_to_bins[(uint8_t)b].push_back((uint8_t)b);
    } else {
        g_log.info("Optimize index bin layout");
        std::vector<std::pair<uint8_t, uint64_t>> sorted(meta.hashes_per_bin.begin(), meta.hashes_per_bin.end());
        std::stable_sort(sorted.begin(), sorted.end(), [](const std::pair<uint8_t, uint64_t> &l, const std::pair<uint8_t, uint64_t> &r) { return l.second < r.second; });
        const uint64_t max_num_hashes = sorted.back().second / 2;
        uint8_t next_bin = 0;
        std::map<std::string, uint8_t> last_bin;
        std::map<uint8_t, uint8_t> bin_to_bucket;
        std::map<uint8_t, uint64_t> new_hashes, new_records;
        for (const auto &pr : sorted) {
            const uint8_t bin = pr.first;
            const std::string &category = meta.bin_to_category.at(bin);
            uint8_t assigned = next_bin;
            auto it = last_bin.find(category);
            if (it != last_bin.end()) {
                if (new_hashes[it->second] + pr.second < max_num_hashes) assigned = it->second; else next_bin++;
            } else {
                next_bin++;
            }
            last_bin[category] = assigned;
            bin_to_bucket[bin] = assigned;
            bucket_to_bins[assigned].push_back(bin);
            new_hashes[assigned] += pr.second;
            new_records[assigned] += meta.records_per_bin.at(bin);
        }
        meta.hashes_per_bin = new_hashes; meta.records_per_bin = new_records;
        std::map<uint8_t, std::string> b2c;
        for (auto &fb : meta.filepath_to_bin) { const uint8_t bucket = bin_to_bucket[fb.second]; b2c[bucket] = meta.bin_to_category.at(fb.second); fb.second = bucket; }
        meta.bin_to_category = b2c;
        meta.num_bins = next_bin;
    }

    // build_index (:238-263)
    uint64_t max_hashes = 0;
    for (const auto &kv : meta.hashes_per_bin) max_hashes = std::max(max_hashes, kv.second);
    const uint64_t S = bin_size_in_bits(opt, max_hashes);
    // no reference produced a single minimiser: seqan3's IBF constructor rejects a bin size of 0 ("The size of a bin must be > 0"),
    // which ends the reference with an uncaught exception; fail as loudly here
    if (S == 0) throw std::runtime_error("no minimisers in any input file: the IBF would have a bin size of 0");
    g_log.info("Create new IBF with " + std::to_string(meta.num_bins) + " bins and " + std::to_string(S) + " bits");
    meta.bins = meta.num_bins; meta.bin_words = (meta.bins + 63) / 64; meta.technical_bins = meta.bin_words * 64;
    meta.bin_size = S; meta.hash_shift = (uint64_t)__builtin_clzll(S); meta.hash_funs = opt.num_hash;
    chn_index_desc d;
    std::memset(&d, 0, sizeof d);
    d.struct_size = sizeof d; d.device = opt.device; d.kmer_size = opt.kmer_size; d.window_size = opt.window_size; d.hash_funs = opt.num_hash;
    d.num_categories = (uint8_t)meta.categories.size(); d.host_index = meta.host_category_index(); d.minimiser_seed = 0x8F3F73B5CF1C9ADEULL;
    d.bins = meta.bins; d.technical_bins = meta.technical_bins; d.bin_size = S; d.hash_shift = meta.hash_shift; d.bin_words = meta.bin_words;
    for (uint64_t b = 0; b < meta.bins; ++b) d.bin_to_category[b] = meta.category_index(meta.bin_to_category.at((uint8_t)b));
    chn_index *index = nullptr;
    CHN_CHECK(chn_index_create(&d, &index));
    for (const auto &kv : bucket_to_bins)
        for (uint8_t bin : kv.second) {
            CHN_CHECK(chn_index_emplace(index, hashes[bin].data(), hashes[bin].size(), kv.first));
            std::vector<uint64_t>().swap(hashes[bin]);
        }

    // Index(...) compresses the IBF (include/index.hpp:43-50) and store_index writes it (include/store_index.hpp:12-17)
    const uint64_t W = meta.bin_words, block_rows = std::max<uint64_t>(1, (256ULL << 20) / (8 * W));
    std::vector<uint64_t> block(block_rows * W);
    uint64_t ones = 0;
    for (uint64_t r0 = 0; r0 < S; r0 += block_rows) {
        const uint64_t nr = std::min(block_rows, S - r0);
        CHN_CHECK(chn_index_download_rows(index, r0, nr, block.data()));
        for (uint64_t i = 0; i < nr * W; ++i) ones += (uint64_t)__builtin_popcountll(block[i]);
    }
    EliasFanoWriter ef;
    ef.begin(meta.technical_bins * S, ones);
    for (uint64_t r0 = 0; r0 < S; r0 += block_rows) {
        const uint64_t nr = std::min(block_rows, S - r0);
        CHN_CHECK(chn_index_download_rows(index, r0, nr, block.data()));
        for (uint64_t i = 0; i < nr * W; ++i) {
            uint64_t x = block[i];
            const uint64_t base = (r0 * W + i) * 64;
            while (x) { ef.add(base + (uint64_t)__builtin_ctzll(x)); x &= x - 1; }
        }
    }
    chn_index_destroy(index);
    g_log.info("Saving index to file " + opt.prefix);
    BinWriter w(opt.prefix);
    w.pod<uint8_t>(meta.window_size); w.pod<uint8_t>(meta.kmer_size); w.pod<double>(meta.max_fpr);
    w.pod<uint8_t>(meta.num_bins);
    w.pod<uint64_t>(meta.categories.size());
    for (const auto &c : meta.categories) w.str(c);
    w.pod<uint64_t>(meta.filepath_to_bin.size());
    for (const auto &fb : meta.filepath_to_bin) { w.str(fb.first); w.pod<uint8_t>(fb.second); }
    w.pod<uint64_t>(meta.bin_to_category.size());
    for (const auto &kv : meta.bin_to_category) { w.pod<uint8_t>(kv.first); w.str(kv.second); }
    w.pod<uint32_t>(meta.num_files);
    w.pod<uint64_t>(meta.records_per_bin.size());
    for (const auto &kv : meta.records_per_bin) { w.pod<uint8_t>(kv.first); w.pod<uint64_t>(kv.second); }
    w.pod<uint64_t>(meta.hashes_per_bin.size());
    for (const auto &kv : meta.hashes_per_bin) { w.pod<uint8_t>(kv.first); w.pod<uint64_t>(kv.second); }
    w.pod<uint64_t>(meta.bins); w.pod<uint64_t>(meta.technical_bins); w.pod<uint64_t>(meta.bin_size);
    w.pod<uint64_t>(meta.hash_shift); w.pod<uint64_t>(meta.bin_words); w.pod<uint64_t>(meta.hash_funs);
    w.pod<uint64_t>(ef.size); w.pod<uint8_t>(ef.wl);
    w.int_vector(ef.wl ? ef.wl : 1, ef.ones * ef.wl, ef.low);
    w.int_vector(1, ef.high_bits, ef.high);
    if (!w.os) throw std::runtime_error("writing " + opt.prefix + " failed");
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) { std::cerr << "A subcommand is required\nRun with --help for more information.\n"; return 106; }
    const std::string sub = argv[1];
    if (sub == "-V" || sub == "--version") { std::cout << CHARON_VERSION << std::endl; return 0; }
    if (sub == "-h" || sub == "--help") {
        std::cout << "Charon: Dehost metagenomic reads\nUsage: charon [OPTIONS] SUBCOMMAND\n\nOptions:\n  -h,--help   Print this help message and exit\n  -V,--version   Show version\n\n"
                     "Subcommands:\n  index    Build an index (IBF) for a number of references split into a small number of bins.\n"
                     "  dehost   Dehost read file into host and other using index.\n"
                     "  (classify is not part of the MI355X hot-path build)\n";
        return 0;
    }
    if (sub == "_records") {  // hidden diagnostic: dump what the block reader sees (no GPU involved)
        try {
            if (argc < 3) return 2;
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
    if (sub != "dehost") { std::cerr << "The following argument was not expected: " << sub << "\nRun with --help for more information.\n"; return 109; }
    DehostArguments opt;
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
