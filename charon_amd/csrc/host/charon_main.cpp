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
// Not implemented: --extract output files (the flag still drives the training cache exactly as in
// the reference), gamma/beta distributions (rejected like an unknown --dist), .bz2 input.
#include <algorithm>
#include <cerrno>
#include <climits>
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
        return s;
    }
    void int_vector(uint8_t &width, uint64_t &bits, std::vector<uint64_t> &words) {
        const long long at = (long long)is_.tellg();
        width = pod<uint8_t>();
        const float growth = pod<float>();
        const uint64_t n_words = pod<uint64_t>();
        bits = pod<uint64_t>();
        if (width < 1 || width > 64 || growth != 1.5f || n_words * 64 < bits)
            throw std::runtime_error("sdsl int_vector framing check failed at file offset " + std::to_string(at) +
                                     " (width/growth_factor/word count/bit size do not agree)");
        words.assign(n_words + 1, 0);
        is_.read(reinterpret_cast<char *>(words.data()), (std::streamsize)(n_words * 8));
        if (!is_) throw std::runtime_error("index file truncated inside an int_vector");
    }

public:
    IndexMeta meta;
    uint64_t ef_size = 0, ef_ones = 0, high_bits = 0;
    uint8_t ef_wl = 0;
    std::vector<uint64_t> low, high;

    explicit IndexFile(const std::string &path) : is_(path, std::ios::binary), path_(path) {
        if (!is_) throw std::runtime_error("cannot open index file " + path);
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

    uint64_t low_at(uint64_t i) const {
        const uint64_t bit = i * ef_wl, wd = bit >> 6, sh = bit & 63;
        uint64_t v = low[wd] >> sh;
        if (sh + ef_wl > 64) v |= low[wd + 1] << (64 - sh);
        return ef_wl == 64 ? v : (v & ((1ULL << ef_wl) - 1));
    }

    // Decode the Elias-Fano vector into plain interleaved rows, `block_rows` at a time (host RAM never holds the
    // whole plain index), and hand each block to `sink(row_begin, n_rows, words)`.
    template <class Sink> void stream_rows(uint64_t block_rows, Sink sink) {
        const IndexMeta &m = meta;
        const uint64_t W = m.bin_words, TB = m.technical_bins, S = m.bin_size;
        std::vector<uint64_t> block(block_rows * W);
        uint64_t row0 = 0, k = 0, hp = 0, prev = 0;
        bool first = true;
        const uint64_t tail_mask = (m.bins & 63) ? ~((1ULL << (m.bins & 63)) - 1) : 0ULL;  // technical bins >= B in the last word
        while (row0 < S) {
            const uint64_t nrows = std::min(block_rows, S - row0);
            std::fill(block.begin(), block.begin() + nrows * W, 0ULL);
            const uint64_t bit_end = (row0 + nrows) * TB;
            while (k < ef_ones) {
                // advance to the next one in m_high
                while (hp < high_bits && !((high[hp >> 6] >> (hp & 63)) & 1)) {
                    const uint64_t rest = high[hp >> 6] >> (hp & 63);
                    hp += rest ? (uint64_t)__builtin_ctzll(rest) : 64 - (hp & 63);
                }
                if (hp >= high_bits) throw std::runtime_error("sd_vector: m_high holds fewer ones than m_low has elements");
                const uint64_t pos = ((hp - k) << ef_wl) | (ef_wl ? low_at(k) : 0);
                if (!first && pos <= prev) throw std::runtime_error("sd_vector: decoded positions are not strictly increasing");
                if (pos >= ef_size) throw std::runtime_error("sd_vector: decoded position beyond m_size");
                if (pos >= bit_end) break;
                const uint64_t wd = pos >> 6;
                if (tail_mask && (wd % W) == W - 1 && ((1ULL << (pos & 63)) & tail_mask))
                    throw std::runtime_error("index has a set bit in a technical bin >= num_bins");
                block[wd - row0 * W] |= 1ULL << (pos & 63);
                prev = pos; first = false;
                ++k; ++hp;
            }
            sink(row0, nrows, block.data());
            row0 += nrows;
        }
        if (k != ef_ones) throw std::runtime_error("sd_vector: not all ones were consumed");
    }
};

// ---------------------------------------------------------------------------------------------------
// FASTA / FASTQ streaming reader (seqan3::sequence_file_input<my_traits> semantics, include/utils.hpp:17-19):
// format by extension, optional .gz, dna5 alphabet (IUPAC -> N, lower case accepted, anything else is an error)
// ---------------------------------------------------------------------------------------------------
struct Record { std::string id, seq, qual; };

class FastxReader {
    gzFile f_ = nullptr;
    bool fastq_ = false;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false;
    std::string pending_;  // header line read ahead (FASTA)
    bool have_pending_ = false;

    bool fill() {
        if (eof_) return false;
        int n = gzread(f_, buf_.data(), (unsigned)buf_.size());
        if (n <= 0) { eof_ = true; len_ = pos_ = 0; return false; }
        len_ = (size_t)n; pos_ = 0;
        return true;
    }
    bool getline(std::string &line) {
        line.clear();
        bool got = false;
        for (;;) {
            if (pos_ >= len_ && !fill()) break;
            got = true;
            const char *b = buf_.data() + pos_;
            const char *nl = static_cast<const char *>(std::memchr(b, '\n', len_ - pos_));
            if (nl) { line.append(b, nl - b); pos_ += (size_t)(nl - b) + 1; break; }
            line.append(b, len_ - pos_);
            pos_ = len_;
        }
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return got;
    }
    static bool valid_dna(char c) { return c && std::strchr("ACGTUNRYSWKMBDHVacgtunryswkmbdhv", c) != nullptr; }

public:
    explicit FastxReader(const std::string &path) : buf_(1 << 20) {
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
    ~FastxReader() { if (f_) gzclose(f_); }
    FastxReader(const FastxReader &) = delete;
    FastxReader &operator=(const FastxReader &) = delete;

    bool next(Record &r) {
        r.id.clear(); r.seq.clear(); r.qual.clear();
        std::string line;
        if (fastq_) {
            do { if (!getline(line)) return false; } while (line.empty());
            if (line[0] != '@') throw std::runtime_error("FASTQ parse error: record does not start with '@'");
            r.id = line.substr(1);
            for (;;) {
                if (!getline(line)) throw std::runtime_error("FASTQ parse error: unexpected end of file");
                if (!line.empty() && line[0] == '+') break;
                r.seq += line;
            }
            while (r.qual.size() < r.seq.size()) {
                if (!getline(line)) throw std::runtime_error("FASTQ parse error: qualities shorter than sequence");
                r.qual += line;
            }
        } else {
            if (!have_pending_) {
                do { if (!getline(pending_)) return false; } while (pending_.empty());
                if (pending_[0] != '>' && pending_[0] != ';') throw std::runtime_error("FASTA parse error: record does not start with '>'");
            }
            r.id = pending_.substr(1);
            have_pending_ = false;
            while (getline(line)) {
                if (!line.empty() && (line[0] == '>' || line[0] == ';')) { pending_ = line; have_pending_ = true; break; }
                for (char c : line) if (!(c == ' ' || c == '\t' || (c >= '0' && c <= '9'))) r.seq.push_back(c);
            }
        }
        for (char c : r.seq) if (!valid_dna(c)) throw std::runtime_error(std::string("parse error: illegal character '") + c + "' in sequence of " + r.id);
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

// get_compression_ratio (src/utils.cpp:114-124) of sequence_to_string(seq) (upper-case dna5 letters, :105-112)
float compression_ratio(const std::string &s1, const std::string &s2) {
    std::string up;
    up.reserve(s1.size() + s2.size());
    for (char c : s1) up.push_back("ACGTN"[dna_code(c)]);
    for (char c : s2) up.push_back("ACGTN"[dna_code(c)]);
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
    std::vector<unsigned char> out(deflateBound(&zs, (uLong)up.size()) + 64);
    zs.next_in = (Bytef *)up.data(); zs.avail_in = (uInt)up.size();
    zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    const size_t compressed = out.size() - zs.avail_out;
    deflateEnd(&zs);
    return static_cast<float>(static_cast<double>(compressed) / static_cast<double>(up.size()));
}

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
                e.call = call[i]; e.conf = conf[i]; e.model_version = training_.version;
            }
        }
    }
    void print(const Entry &e) {  // print_assignment_result, include/read_entry.hpp:322-337
        out_ << (e.call == 255 ? "U" : "C") << "\t";
        out_.precision(6);
        out_ << e.read_id << "\t" << meta_.category_name(e.call) << "\t" << e.length << "\t" << e.num_hashes << "\t" << e.mean_quality << "\t"
             << +e.conf << "\t" << e.compression << "\t";
        for (size_t i = 0; i < meta_.categories.size(); ++i) {
            const float prop = static_cast<float>(e.counts[i]) / static_cast<float>(e.num_hashes);      // get_proportions :140-150
            const float uprop = static_cast<float>(e.unique[i]) / static_cast<float>(e.num_hashes);
            out_ << meta_.categories[i] << ":" << e.counts[i] << ":" << prop << ":" << uprop << ":" << e.prob[i] << " ";
        }
        out_ << "\n";
    }
    void classify_read(Entry &e) {  // include/result.hpp:97-116
        if (e.model_version != training_.version) { std::vector<Entry *> one(1, &e); reclassify(one); }
        print(e);
        if (e.call < 255) classified_counts_[e.call] += 1; else unclassified_ += 1;
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
    }
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
    std::vector<Record> r1, r2;
    std::vector<uint32_t> bases, nmask, len1, len2;
    std::vector<uint64_t> off1, off2;
    std::vector<float> mq, comp;
    bool any_n = false;
    uint64_t n_bases = 0;

    void clear() { r1.clear(); r2.clear(); }
    static uint64_t pad64(uint64_t x) { return (x + 63) & ~63ULL; }
    void put(const std::string &s, uint64_t off) {
        for (size_t i = 0; i < s.size(); ++i) {
            const int c = dna_code(s[i]);
            const uint64_t j = off + i;
            if (c == 4) { nmask[j >> 5] |= 1u << (j & 31); any_n = true; }
            else bases[j >> 4] |= (uint32_t)c << (2 * (j & 15));
        }
    }
    void pack(bool paired, int threads) {
        const size_t n = r1.size();
        off1.assign(n, 0); len1.assign(n, 0); mq.assign(n, 0); comp.assign(n, 0);
        if (paired) { off2.assign(n, 0); len2.assign(n, 0); }
        uint64_t cur = 0;
        for (size_t i = 0; i < n; ++i) {
            off1[i] = cur; len1[i] = (uint32_t)r1[i].seq.size(); cur += pad64(len1[i]);
            if (paired) { off2[i] = cur; len2[i] = (uint32_t)r2[i].seq.size(); cur += pad64(len2[i]); }
        }
        n_bases = std::max<uint64_t>(cur, 64);
        bases.assign(n_bases / 16, 0); nmask.assign(n_bases / 32, 0); any_n = false;
        for (size_t i = 0; i < n; ++i) { put(r1[i].seq, off1[i]); if (paired) put(r2[i].seq, off2[i]); }
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
        for (long i = 0; i < (long)n; ++i) {
            // mean quality (src/dehost_main.cpp:355-360 / :441-450): int sum of phred (char - 33) / count, as float
            int sum = 0;
            size_t cnt = r1[i].qual.size();
            for (char c : r1[i].qual) sum += (int)c - 33;
            if (paired) { cnt += r2[i].qual.size(); for (char c : r2[i].qual) sum += (int)c - 33; }
            mq[i] = cnt ? static_cast<float>(sum) / static_cast<float>(cnt) : 0.0f;
            if (len1[i] + (paired ? len2[i] : 0u) > 0) comp[i] = compression_ratio(r1[i].seq, paired ? r2[i].seq : std::string());
        }
    }
};

std::string first_token(const std::string &id) {  // split(id, " ")[0] (src/utils.cpp:9-20)
    const size_t e = id.find(' ');
    return e == std::string::npos ? id : id.substr(0, e);
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
    if (opt.run_extract) g_log.warn("--extract: output files are not written by this build; the flag only drives the training cache");
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
    const uint64_t block_rows = std::max<uint64_t>(1, (256ULL << 20) / (8 * meta.bin_words));
    file.stream_rows(block_rows, [&](uint64_t row0, uint64_t nrows, const uint64_t *words) { CHN_CHECK(chn_index_upload_rows(index, row0, nrows, words)); });
    file.low.clear(); file.low.shrink_to_fit(); file.high.clear(); file.high.shrink_to_fit();
    g_log.info("Index loaded");

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

    std::ios::sync_with_stdio(false);
    Result result(meta, opt, stream, model, std::cout);
    g_log.info("Dehosting file " + opt.read_file + (opt.is_paired ? " and " + opt.read_file2 : ""));

    FastxReader in1(opt.read_file);
    std::unique_ptr<FastxReader> in2;
    if (opt.is_paired) in2.reset(new FastxReader(opt.read_file2));
    const size_t C = meta.categories.size();
    HostBatch hb;
    std::vector<uint32_t> nh, cnt, unq;
    std::vector<double> prob;
    std::vector<uint8_t> call, conf, flags;
    bool more = true, have_carry = false;
    Record a, b;
    while (more) {
        hb.clear();
        uint64_t bases = 0;
        while (hb.r1.size() < opt.batch_reads) {
            if (have_carry) {  // a read pair that did not fit the previous batch
                have_carry = false;
            } else {
            if (!in1.next(a)) { more = false; break; }
            if (opt.is_paired) {
                if (!in2->next(b)) { more = false; break; }  // the second file is simply `take`n (src/dehost_main.cpp:413-415)
                std::string id1 = a.id, id2 = b.id;
                if (!id1.empty()) id1.erase(id1.size() - 1);
                if (!id2.empty()) id2.erase(id2.size() - 1);
                if (id1 != id2) {  // :423-430: prints to stdout and throws inside the OpenMP region -> terminate
                    std::cout << id1 << " " << id2;
                    std::cout.flush();
                    std::fprintf(stderr, "terminate called after throwing an instance of 'std::runtime_error'\n  what():  Your pairs don't match for read ids.\n");
                    std::abort();
                }
            }
            const uint64_t L = a.seq.size() + (opt.is_paired ? b.seq.size() : 0);
            if (L == 0) { g_log.warn("Ignoring read " + a.id + " as has zero length!"); continue; }  // :351-354
            if (L > std::numeric_limits<uint32_t>::max()) { g_log.warn("Ignoring read " + a.id + " as too long!"); continue; }
            const uint64_t need = HostBatch::pad64(a.seq.size()) + (opt.is_paired ? HostBatch::pad64(b.seq.size()) : 0);
            if (need > opt.batch_bases) throw std::runtime_error("read " + a.id + " is longer than CHARON_BATCH_BASES");
            }
            const uint64_t need = HostBatch::pad64(a.seq.size()) + (opt.is_paired ? HostBatch::pad64(b.seq.size()) : 0);
            if (bases + need > opt.batch_bases) { have_carry = true; break; }
            hb.r1.push_back(std::move(a));
            if (opt.is_paired) hb.r2.push_back(std::move(b));
            bases += need;
        }
        const size_t n = hb.r1.size();
        if (n == 0) continue;
        hb.pack(opt.is_paired, opt.threads);
        result.ensure_device_model();
        const uint32_t version = result.current_model_version();
        chn_batch bt;
        std::memset(&bt, 0, sizeof bt);
        bt.struct_size = sizeof bt; bt.on_device = 0; bt.n_reads = n; bt.n_bases = hb.n_bases;
        bt.bases2 = hb.bases.data(); bt.nmask = hb.any_n ? hb.nmask.data() : nullptr;
        bt.seg1_offset = hb.off1.data(); bt.seg1_length = hb.len1.data();
        bt.seg2_offset = opt.is_paired ? hb.off2.data() : nullptr; bt.seg2_length = opt.is_paired ? hb.len2.data() : nullptr;
        bt.mean_quality = hb.mq.data(); bt.compression = hb.comp.data();
        CHN_CHECK(chn_batch_submit(stream, &bt));
        nh.resize(n); cnt.resize(n * C); unq.resize(n * C); prob.resize(n * C); call.resize(n); conf.resize(n); flags.resize(n);
        chn_result rs;
        std::memset(&rs, 0, sizeof rs);
        rs.struct_size = sizeof rs; rs.on_device = 0;
        rs.num_hashes = nh.data(); rs.counts = cnt.data(); rs.unique_counts = unq.data(); rs.probabilities = prob.data();
        rs.call = call.data(); rs.confidence = conf.data(); rs.flags = flags.data();
        CHN_CHECK(chn_batch_wait(stream, &rs));
        // critical(add_read_to_results): serial, in input order (what the reference does at -t 1)
        for (size_t i = 0; i < n; ++i) {
            Entry e;
            e.read_id = first_token(hb.r1[i].id);
            e.length = hb.len1[i] + (opt.is_paired ? hb.len2[i] : 0u);
            e.num_hashes = nh[i]; e.mean_quality = hb.mq[i]; e.compression = hb.comp[i];
            e.counts.assign(cnt.begin() + i * C, cnt.begin() + (i + 1) * C);
            e.unique.assign(unq.begin() + i * C, unq.begin() + (i + 1) * C);
            e.prob.assign(prob.begin() + i * C, prob.begin() + (i + 1) * C);
            e.call = call[i]; e.conf = conf[i]; e.model_version = version;
            result.add_read(e);
        }
    }
    result.complete();
    std::cout.flush();
    result.print_summary();
    chn_stream_destroy(stream);
    chn_index_destroy(index);
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) { std::cerr << "A subcommand is required\nRun with --help for more information.\n"; return 106; }
    const std::string sub = argv[1];
    if (sub == "-V" || sub == "--version") { std::cout << CHARON_VERSION << std::endl; return 0; }
    if (sub == "-h" || sub == "--help") {
        std::cout << "Charon: Dehost metagenomic reads\nUsage: charon [OPTIONS] SUBCOMMAND\n\nOptions:\n  -h,--help   Print this help message and exit\n  -V,--version   Show version\n\n"
                     "Subcommands:\n  dehost   Dehost read file into host and other using index.\n"
                     "  (index and classify are not part of the MI355X hot-path build)\n";
        return 0;
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
