// ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (C++14, no dependencies but zlib) of the per-read classification path of
// `charon dehost` (rmcolq/charon @ 2025-07-04).  It exists to CHECK the HIP implementation in
// charon_amd/; nothing in the product path may include, link or execute it.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// PARITY STATUS: "parity unpinned" with respect to real reference output.  The reference ships no
// tests, golden vectors or fixtures, and its arithmetic lives in third-party libraries that are not
// in /root/reference (seqan3 @30bdf8d0 with sdsl-lite v3 + cereal, kthohr/stats 3.4.0, gzip-hpp
// @7546b35 + zlib; cmake/package-lock.cmake:5-57).  What pins this file instead:
//   * public-documentation known-answer vectors of the seqan3 primitives (tests/test_oracle_kat.py),
//   * an independent numpy/python restatement (oracle/pyref.py) that must agree bit-for-bit,
//   * the cross-check values recorded in SURVEY.md App. A.8.
//
// Every function cites the reference file:line (relative to /root/reference) it follows.
#pragma once

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include <zlib.h>

namespace oracle {

typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------------------------
// Alphabet (seqan3::dna5 as selected by my_traits, include/utils.hpp:17-19).
// Ranks A0 C1 G2 N3 T4; every IUPAC letter other than ACGT (either case) folds to N.
// ---------------------------------------------------------------------------------------------
inline uint8_t dna5_rank(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 4;
        default: return 3;
    }
}
inline char dna5_char(uint8_t r) { return "ACGNT"[r]; }
// complement table [4,2,1,3,0] (A<->T, C<->G, N<->N)
inline uint8_t dna5_comp(uint8_t r) { static const uint8_t t[5] = {4, 2, 1, 3, 0}; return t[r]; }

// seqan3 accepts exactly the IUPAC nucleotide letters for a dna5 sequence file field; anything else
// is a parse error.  (seqan3 char_is_valid_for<dna5>.)
inline bool dna5_char_valid(char c) {
    static const char *ok = "ACGTUNRYSWKMBDHVacgtunryswkmbdhv";
    return c != 0 && std::strchr(ok, c) != nullptr;
}

static const uint64_t MINIMISER_SEED = 0x8F3F73B5CF1C9ADEULL;

// ---------------------------------------------------------------------------------------------
// seqan3::views::minimiser_hash(shape{ungapped{k}}, window_size{w}) as used at
// src/dehost_main.cpp:317-318,367.  Generic over the alphabet size so that the public dna4
// documentation vectors can be checked too (sigma = 4: complement = 3 - r).
// ---------------------------------------------------------------------------------------------
struct Alphabet {
    unsigned sigma;
    uint8_t comp[8];
};
inline Alphabet alphabet_dna5() { Alphabet a; a.sigma = 5; a.comp[0] = 4; a.comp[1] = 2; a.comp[2] = 1; a.comp[3] = 3; a.comp[4] = 0; return a; }
inline Alphabet alphabet_dna4() { Alphabet a; a.sigma = 4; a.comp[0] = 3; a.comp[1] = 2; a.comp[2] = 1; a.comp[3] = 0; return a; }

// kmer_hash of every k-mer of `ranks` (most significant digit first), forward strand.
inline std::vector<uint64_t> kmer_hashes(const std::vector<uint8_t> &ranks, unsigned k, const Alphabet &al) {
    std::vector<uint64_t> out;
    if (ranks.size() < k || k == 0) return out;
    uint64_t top = 1;
    for (unsigned i = 1; i < k; ++i) top *= al.sigma;
    uint64_t h = 0;
    for (unsigned i = 0; i < k; ++i) h = h * al.sigma + ranks[i];
    out.push_back(h);
    for (size_t i = k; i < ranks.size(); ++i) {
        h = (h - ranks[i - k] * top) * al.sigma + ranks[i];
        out.push_back(h);
    }
    return out;
}

// canonical values v[i] = min(fwd[i]^seed, rc[i]^seed)
inline std::vector<uint64_t> canonical_values(const std::vector<uint8_t> &ranks, unsigned k, const Alphabet &al,
                                              uint64_t seed = MINIMISER_SEED) {
    std::vector<uint64_t> fwd = kmer_hashes(ranks, k, al);
    std::vector<uint64_t> v(fwd.size());
    if (fwd.empty()) return v;
    // reverse complement strand: hash of revcomp(kmer_i) = sum_m comp(c[i+m]) * sigma^m
    uint64_t top = 1;
    for (unsigned i = 1; i < k; ++i) top *= al.sigma;
    uint64_t rc = 0, p = 1;
    for (unsigned m = 0; m < k; ++m) { rc += al.comp[ranks[m]] * p; p *= al.sigma; }
    v[0] = std::min(fwd[0] ^ seed, rc ^ seed);
    for (size_t i = 1; i < fwd.size(); ++i) {
        rc = (rc - al.comp[ranks[i - 1]]) / al.sigma + al.comp[ranks[i + k - 1]] * top;
        v[i] = std::min(fwd[i] ^ seed, rc ^ seed);
    }
    return v;
}

// The seqan3 minimiser view's emission rule: window of Wn = w-k+1 values,
// rightmost minimum on (re)computation, strict '<' for a new arrival, re-emission when the tracked
// minimum leaves the window even if the value is unchanged.
inline std::vector<uint64_t> minimiser_hash(const std::vector<uint8_t> &ranks, unsigned k, unsigned w,
                                            const Alphabet &al, uint64_t seed = MINIMISER_SEED) {
    std::vector<uint64_t> out;
    std::vector<uint64_t> v = canonical_values(ranks, k, al, seed);
    if (v.empty()) return out;
    size_t Wn = (w >= k) ? (w - k + 1) : 1;
    size_t wl = std::min(Wn, v.size());
    auto rightmost_min = [&](size_t begin, size_t len, uint64_t &mv, size_t &off) {
        mv = v[begin]; off = 0;
        for (size_t t = 1; t < len; ++t)
            if (v[begin + t] <= mv) { mv = v[begin + t]; off = t; }
    };
    uint64_t mv; size_t off;
    rightmost_min(0, wl, mv, off);
    out.push_back(mv);
    for (size_t j = wl; j < v.size(); ++j) {
        uint64_t x = v[j];
        if (off == 0) {
            rightmost_min(j - Wn + 1, Wn, mv, off);
            out.push_back(mv);
        } else if (x < mv) {
            mv = x; off = Wn - 1;
            out.push_back(mv);
        } else {
            --off;
        }
    }
    return out;
}

inline std::vector<uint8_t> ranks_of(const std::string &s) {
    std::vector<uint8_t> r(s.size());
    for (size_t i = 0; i < s.size(); ++i) r[i] = dna5_rank(s[i]);
    return r;
}

// ---------------------------------------------------------------------------------------------
// seqan3::interleaved_bloom_filter addressing; used by Index::agent()
// (include/index.hpp:110-112) and bulk_contains at src/dehost_main.cpp:368.
// ---------------------------------------------------------------------------------------------
static const uint64_t IBF_SEEDS[5] = {13572355802537770549ULL, 13043817825332782213ULL, 10650232656628343401ULL,
                                      16499269484942379435ULL, 4893150838803335377ULL};

inline unsigned clz64(uint64_t x) { return x ? (unsigned)__builtin_clzll(x) : 64u; }

inline uint64_t hash_and_fit_row(uint64_t x, uint64_t seed, uint64_t bin_size, unsigned hash_shift) {
    x *= seed;
    x ^= x >> hash_shift;
    x *= 11400714819323198485ULL;
    return (uint64_t)(((u128)x * (u128)bin_size) >> 64);  // fastrange
}

// ---------------------------------------------------------------------------------------------
// sdsl::int_vector<0> and sdsl::select_support_mcl<t_b, 1> (sdsl-lite v3, bundled with seqan3 @30bdf8d0; the
// source is NOT in the image: [3P-recall], parity unpinned).  An sd_vector ends with two of them over m_high
// (m_high_1_select, m_high_0_select) and upstream charon's archive(ibf_) (include/index.hpp:130) stores both,
// so a file upstream can load needs them.  The construction below restates select_support_mcl.hpp's
// initData / init_slow / init_fast STATEMENT BY STATEMENT (the product writer derives its rules from the
// outcome instead: charon_amd/csrc/host/sdsl_select.inc; the two must agree byte for byte).
// ---------------------------------------------------------------------------------------------
struct IntVec {  // int_vector<0>: `n` entries of `width` bits, packed LSB first into 64-bit words
    uint8_t width = 64;  // a default-constructed int_vector<0> has width 64 and no entries
    uint64_t n = 0;
    std::vector<uint64_t> words;
    IntVec() {}
    IntVec(uint64_t size, uint64_t def, uint8_t w) : width(w), n(size), words((size * w + 63) / 64 + 1, 0) { (void)def; }
    bool empty() const { return n == 0; }
    uint64_t bit_size() const { return n * width; }
    uint64_t get(uint64_t i) const {
        const uint64_t bit = i * width, wd = bit >> 6, sh = bit & 63;
        uint64_t v = words[wd] >> sh;
        if (sh + width > 64) v |= words[wd + 1] << (64 - sh);
        return width == 64 ? v : (v & ((1ULL << width) - 1));
    }
    void set(uint64_t i, uint64_t v) {  // (entries are written once, onto zeros; the value is cut to `width` bits as sdsl does)
        if (width < 64) v &= (1ULL << width) - 1;
        const uint64_t bit = i * width, wd = bit >> 6, sh = bit & 63;
        words[wd] |= v << sh;
        if (sh + width > 64) words[wd + 1] |= v >> (64 - sh);
    }
};

struct SelectMcl {
    // the bit vector it supports
    const uint64_t *v_words = nullptr;
    uint64_t v_size = 0;
    int t_b = 1;
    // members, in the order cereal stores them
    uint64_t m_arg_cnt = 0;
    uint32_t m_logn = 0, m_logn2 = 0, m_logn4 = 0;
    IntVec m_superblock;
    std::vector<IntVec> m_longsuperblock;  // empty() <=> nullptr in sdsl
    std::vector<IntVec> m_miniblock;

    static unsigned hi(uint64_t x) { return x ? 63u - clz64(x) : 0u; }               // bits::hi
    static unsigned sel(uint64_t x, unsigned i) { while (--i) x &= x - 1; return (unsigned)__builtin_ctzll(x); }  // bits::sel, i from 1
    bool v_at(uint64_t i) const { return (v_words[i >> 6] >> (i & 63)) & 1; }
    uint64_t v_word(uint64_t i) const { return v_words[i]; }  // padding bits behind v_size are 0, as in an sdsl bit_vector
    // select_support_trait<t_b, 1>
    bool found_arg(uint64_t i) const { return t_b ? v_at(i) : !v_at(i); }
    uint64_t args_in_the_word(uint64_t w) const { return (uint64_t)__builtin_popcountll(t_b ? w : ~w); }
    unsigned ith_arg_pos_in_the_word(uint64_t w, uint64_t i) const { return sel(t_b ? w : ~w, (unsigned)i); }
    uint64_t arg_cnt() const {
        uint64_t ones = 0;
        for (uint64_t i = 0; i < v_size; ++i) ones += v_at(i);
        return t_b ? ones : v_size - ones;
    }

    void initData() {
        m_arg_cnt = 0;
        m_logn = hi(((v_size + 63) >> 6) << 6) + 1;
        m_logn2 = m_logn * m_logn;
        m_logn4 = m_logn2 * m_logn2;
        m_longsuperblock.clear();
        m_miniblock.clear();
    }
    // select_support_mcl(const bit_vector *): init_slow for vectors below 100 000 bits, init_fast otherwise
    void init(const uint64_t *words, uint64_t size, int b) {
        v_words = words; v_size = size; t_b = b;
        if (v_size < 100000) init_slow(); else init_fast();
    }
    void init_slow() {
        initData();
        m_arg_cnt = arg_cnt();
        const uint64_t SUPER_BLOCK_SIZE = 4096;
        if (m_arg_cnt == 0) return;
        const uint64_t sb = (m_arg_cnt + SUPER_BLOCK_SIZE - 1) / SUPER_BLOCK_SIZE;
        m_miniblock.assign(sb, IntVec());
        m_superblock = IntVec(sb, 0, (uint8_t)m_logn);
        std::vector<uint64_t> arg_position(SUPER_BLOCK_SIZE);
        uint64_t cnt = 0, sb_cnt = 0;
        for (uint64_t i = 0; i < v_size; ++i) {
            if (!found_arg(i)) continue;
            arg_position[cnt % SUPER_BLOCK_SIZE] = i;
            ++cnt;
            if (cnt % SUPER_BLOCK_SIZE == 0 || cnt == m_arg_cnt) {
                m_superblock.set(sb_cnt, arg_position[0]);
                const uint64_t pos_diff = arg_position[(cnt - 1) % SUPER_BLOCK_SIZE] - arg_position[0];
                if (pos_diff > m_logn4) {  // longblock
                    if (m_longsuperblock.empty()) m_longsuperblock.assign(sb, IntVec());
                    m_longsuperblock[sb_cnt] = IntVec(SUPER_BLOCK_SIZE, 0, (uint8_t)(hi(arg_position[(cnt - 1) % SUPER_BLOCK_SIZE]) + 1));
                    for (uint64_t j = 0; j <= (cnt - 1) % SUPER_BLOCK_SIZE; ++j) m_longsuperblock[sb_cnt].set(j, arg_position[j]);
                } else {  // short block
                    m_miniblock[sb_cnt] = IntVec(64, 0, (uint8_t)(hi(pos_diff) + 1));
                    for (uint64_t j = 0; j <= (cnt - 1) % SUPER_BLOCK_SIZE; j += 64) m_miniblock[sb_cnt].set(j / 64, arg_position[j] - arg_position[0]);
                }
                ++sb_cnt;
            }
        }
    }
    void init_fast() {
        initData();
        m_arg_cnt = arg_cnt();
        const uint64_t SUPER_BLOCK_SIZE = 64 * 64;
        if (m_arg_cnt == 0) return;
        const uint64_t sb = (m_arg_cnt + SUPER_BLOCK_SIZE - 1) / SUPER_BLOCK_SIZE;
        m_miniblock.assign(sb, IntVec());
        m_superblock = IntVec(sb, 0, (uint8_t)m_logn);
        std::vector<uint64_t> arg_position(SUPER_BLOCK_SIZE);
        uint64_t last_k64 = 1, sb_cnt = 0;
        uint64_t cnt_old = 0, cnt_new = 0, last_k64_sum = 1;
        for (uint64_t i = 0, wi = 0; i < (((v_size + 63) >> 6) << 6); i += 64, ++wi) {
            const uint64_t data = v_word(wi);
            cnt_new += args_in_the_word(data);
            cnt_new = std::min(cnt_new, m_arg_cnt);  // zeros in the padding behind the vector are no args
            if (cnt_new >= last_k64_sum) {
                arg_position[last_k64 - 1] = i + ith_arg_pos_in_the_word(data, last_k64_sum - cnt_old);
                last_k64 += 64;
                last_k64_sum += 64;
                if (last_k64 == SUPER_BLOCK_SIZE + 1) {
                    m_superblock.set(sb_cnt, arg_position[0]);
                    uint64_t pos_of_last_arg_in_the_block = arg_position[last_k64 - 65];
                    for (uint64_t ii = arg_position[last_k64 - 65] + 1, j = last_k64 - 65; ii < v_size && j < SUPER_BLOCK_SIZE; ++ii)
                        if (found_arg(ii)) { pos_of_last_arg_in_the_block = ii; ++j; }
                    const uint64_t pos_diff = pos_of_last_arg_in_the_block - arg_position[0];
                    if (pos_diff > m_logn4) {  // long block
                        if (m_longsuperblock.empty()) m_longsuperblock.assign(sb + 1, IntVec());
                        m_longsuperblock[sb_cnt] = IntVec(SUPER_BLOCK_SIZE, 0, (uint8_t)(hi(pos_of_last_arg_in_the_block) + 1));
                        for (uint64_t j = arg_position[0], k = 0; k < SUPER_BLOCK_SIZE && j <= pos_of_last_arg_in_the_block; ++j)
                            if (found_arg(j)) m_longsuperblock[sb_cnt].set(k++, j);
                    } else {
                        m_miniblock[sb_cnt] = IntVec(64, 0, (uint8_t)(hi(pos_diff) + 1));
                        for (uint64_t j = 0; j < SUPER_BLOCK_SIZE; j += 64) m_miniblock[sb_cnt].set(j / 64, arg_position[j] - arg_position[0]);
                    }
                    ++sb_cnt;
                    last_k64 = 1;
                }
            }
            cnt_old = cnt_new;
        }
        if (last_k64 > 1) {  // handle last block: append long superblock
            if (m_longsuperblock.empty()) m_longsuperblock.assign(sb + 1, IntVec());
            m_longsuperblock[sb_cnt] = IntVec(SUPER_BLOCK_SIZE, 0, (uint8_t)(hi(v_size - 1) + 1));
            for (uint64_t i = arg_position[0], k = 0; i < v_size; ++i)
                if (found_arg(i)) m_longsuperblock[sb_cnt].set(k++, i);
            ++sb_cnt;
        }
    }
    // select_support_mcl::select(i), i from 1 -- reads only the stored blocks and, between two sampled args, the bit vector
    uint64_t select(uint64_t i) const {
        i = i - 1;
        const uint64_t sb_idx = i >> 12, offset = i & 0xFFF;
        if (!m_longsuperblock.empty() && !m_longsuperblock[sb_idx].empty()) return m_longsuperblock[sb_idx].get(offset);
        if ((offset & 0x3F) == 0) return m_superblock.get(sb_idx) + m_miniblock[sb_idx].get(offset >> 6);
        i = i - (sb_idx << 12) - ((offset >> 6) << 6);
        uint64_t pos = m_superblock.get(sb_idx) + m_miniblock[sb_idx].get(offset >> 6) + 1;
        for (;; ++pos)
            if (found_arg(pos) && --i == 0) return pos;
    }
};

// Elias-Fano bit vector = sdsl::sd_vector<> (storage of the *compressed* IBF the reference keeps in
// RAM, include/index.hpp:26).  get_int() is what every reference probe costs.
struct SdVector {
    uint64_t size = 0;          // m_size: length of the plain bit vector
    uint8_t wl = 0;             // width of the low parts
    uint64_t ones = 0;          // number of set bits (m_low element count)
    std::vector<uint64_t> low;  // packed, `wl` bits per element
    std::vector<uint64_t> high; // bit vector, length ones + 2^logm
    uint64_t high_bits = 0;
    // select_0 samples over `high`: position of every 2^SAMPLE_LOG-th zero
    static const unsigned SAMPLE_LOG = 9;
    std::vector<uint64_t> sel0_samples;
    // the file's own select structures (load_index fills them when the file carries them; the probes above do not use them)
    SelectMcl sel1, sel0;
    bool has_sel = false;

    static unsigned hi(uint64_t x) { return x ? 63u - clz64(x) : 0u; }

    uint64_t low_at(uint64_t i) const {
        if (wl == 0) return 0;
        uint64_t bit = i * wl, wd = bit >> 6, sh = bit & 63;
        uint64_t val = low[wd] >> sh;
        if (sh + wl > 64) val |= low[wd + 1] << (64 - sh);
        return wl == 64 ? val : (val & ((1ULL << wl) - 1));
    }
    void low_set(uint64_t i, uint64_t val) {
        if (wl == 0) return;
        uint64_t bit = i * wl, wd = bit >> 6, sh = bit & 63;
        low[wd] |= val << sh;
        if (sh + wl > 64) low[wd + 1] |= val >> (64 - sh);
    }

    // sdsl sd_vector construction rule
    template <class It> void build(It begin, It end, uint64_t universe) {
        size = universe;
        ones = (uint64_t)std::distance(begin, end);
        unsigned logm = hi(ones) + 1, logn = hi(size) + 1;
        if (logm == logn) --logm;
        wl = (uint8_t)(logn - logm);
        high_bits = ones + (1ULL << logm);
        low.assign((ones * wl + 63) / 64 + 1, 0);
        high.assign((high_bits + 63) / 64, 0);
        uint64_t k = 0;
        for (It it = begin; it != end; ++it, ++k) {
            uint64_t pos = *it;
            low_set(k, wl == 64 ? pos : (pos & ((wl ? (1ULL << wl) : 1ULL) - 1)));
            uint64_t hp = (wl >= 64 ? 0 : (pos >> wl)) + k;
            high[hp >> 6] |= 1ULL << (hp & 63);
        }
        build_select();
    }
    void build_select() {  // word-wise: the vector of a 39 GB index has ~7e10 zeros
        sel0_samples.clear();
        const uint64_t step = 1ULL << SAMPLE_LOG;
        uint64_t zeros = 0;
        for (uint64_t wd = 0; wd < high.size(); ++wd) {
            uint64_t z = ~high[wd];
            if (wd == high.size() - 1 && (high_bits & 63)) z &= (1ULL << (high_bits & 63)) - 1;
            const uint64_t c = (uint64_t)__builtin_popcountll(z);
            // zeros numbered zeros .. zeros+c-1 live in this word; the sampled ones are the multiples of `step`
            uint64_t next = (zeros + step - 1) & ~(step - 1);
            while (c && next < zeros + c) {
                uint64_t t = z;
                for (uint64_t skip = next - zeros; skip; --skip) t &= t - 1;
                sel0_samples.push_back(wd * 64 + (unsigned)__builtin_ctzll(t));
                next += step;
            }
            zeros += c;
        }
    }
    // Same construction straight from the plain interleaved words, in parallel and without materialising the list of set
    // positions (a 39 GB index holds 5e10 of them).  Threads own disjoint ranges of ones; words of `low` / `high` at range
    // borders are shared, hence the atomic ORs.
    void build_from_words(const std::vector<uint64_t> &plain, uint64_t universe) {
        const uint64_t CH = 1ULL << 16, n = plain.size(), nch = (n + CH - 1) / CH;
        std::vector<uint64_t> pre(nch + 1, 0);
#pragma omp parallel for schedule(static)
        for (long c = 0; c < (long)nch; ++c) {
            uint64_t cnt = 0;
            const uint64_t hi_w = std::min(n, ((uint64_t)c + 1) * CH);
            for (uint64_t wd = (uint64_t)c * CH; wd < hi_w; ++wd) cnt += (uint64_t)__builtin_popcountll(plain[wd]);
            pre[c + 1] = cnt;
        }
        for (uint64_t c = 0; c < nch; ++c) pre[c + 1] += pre[c];
        size = universe;
        ones = pre[nch];
        unsigned logm = hi(ones) + 1, logn = hi(size) + 1;
        if (logm == logn) --logm;
        wl = (uint8_t)(logn - logm);
        high_bits = ones + (1ULL << logm);
        low.assign((ones * wl + 63) / 64 + 1, 0);
        high.assign((high_bits + 63) / 64, 0);
        const uint64_t mask = wl >= 64 ? ~0ULL : ((wl ? (1ULL << wl) : 1ULL) - 1);
#pragma omp parallel for schedule(dynamic, 16)
        for (long c = 0; c < (long)nch; ++c) {
            uint64_t k = pre[c];
            const uint64_t hi_w = std::min(n, ((uint64_t)c + 1) * CH);
            for (uint64_t wd = (uint64_t)c * CH; wd < hi_w; ++wd) {
                uint64_t x = plain[wd];
                while (x) {
                    const uint64_t pos = wd * 64 + (unsigned)__builtin_ctzll(x);
                    x &= x - 1;
                    if (wl) {
                        const uint64_t val = pos & mask, bit = k * wl, w0 = bit >> 6, sh = bit & 63;
                        __atomic_fetch_or(&low[w0], val << sh, __ATOMIC_RELAXED);
                        if (sh + wl > 64) __atomic_fetch_or(&low[w0 + 1], val >> (64 - sh), __ATOMIC_RELAXED);
                    }
                    const uint64_t hp = (wl >= 64 ? 0 : (pos >> wl)) + k;
                    __atomic_fetch_or(&high[hp >> 6], 1ULL << (hp & 63), __ATOMIC_RELAXED);
                    ++k;
                }
            }
        }
        build_select();
    }
    // position in `high` of the j-th zero (0-based); returns high_bits if there is none
    uint64_t select0(uint64_t j) const {
        uint64_t s = j >> SAMPLE_LOG;
        if (s >= sel0_samples.size()) return high_bits;
        uint64_t pos = sel0_samples[s];
        uint64_t need = j - (s << SAMPLE_LOG);  // zeros still to skip after `pos` (pos itself is a zero)
        uint64_t wd = pos >> 6;
        uint64_t z = ~high[wd] & (~0ULL << (pos & 63));
        for (;;) {
            if (wd == high.size() - 1 && (high_bits & 63)) z &= (1ULL << (high_bits & 63)) - 1;
            unsigned c = (unsigned)__builtin_popcountll(z);
            if (need < c) {
                for (uint64_t t = 0; t < need; ++t) z &= z - 1;
                return wd * 64 + (unsigned)__builtin_ctzll(z);
            }
            need -= c;
            if (++wd >= high.size()) return high_bits;
            z = ~high[wd];
        }
    }
    // 64 bits of the plain vector starting at bit `idx` (sd_vector::get_int(idx, 64))
    uint64_t get_int(uint64_t idx) const {
        uint64_t result = 0;
        uint64_t end = idx + 64;
        uint64_t bucket = wl >= 64 ? 0 : (idx >> wl);
        // start of bucket `bucket` in high = just after the (bucket-1)-th zero
        uint64_t hp = bucket == 0 ? 0 : select0(bucket - 1) + 1;
        uint64_t k = hp - bucket;  // ones before
        while (hp < high_bits && k < ones) {
            if ((high[hp >> 6] >> (hp & 63)) & 1) {
                uint64_t pos = ((hp - k) << wl) | low_at(k);
                if (pos >= end) break;
                if (pos >= idx) result |= 1ULL << (pos - idx);
                ++k;
            } else {
                if (((hp - k + 1) << wl) >= end) break;  // next bucket starts past the range
            }
            ++hp;
        }
        return result;
    }
    // decode all set positions (ascending)
    template <class F> void for_each_one(F f) const {
        uint64_t k = 0;
        for (uint64_t hp = 0; hp < high_bits && k < ones; ++hp)
            if ((high[hp >> 6] >> (hp & 63)) & 1) { f(((hp - k) << wl) | low_at(k)); ++k; }
    }
};

struct InputSummary {  // include/input_summary.hpp:16-94
    uint8_t num_bins = 0;
    std::vector<std::string> categories;
    std::vector<std::pair<std::string, uint8_t>> filepath_to_bin;
    std::map<uint8_t, std::string> bin_to_category;  // unordered_map in the reference; order irrelevant to readers

    uint8_t num_categories() const { return (uint8_t)categories.size(); }
    uint8_t category_index(const std::string &c) const {  // :39-45
        for (size_t i = 0; i < categories.size(); ++i)
            if (c == categories[i]) return (uint8_t)i;
        return 255;
    }
    uint8_t host_category_index() const {  // :47-55
        return std::min(category_index("human"), category_index("host"));
    }
    std::string category_name(uint8_t index) const {  // :57-62 ('>' not '>=': reproduced)
        if (index > categories.size()) return "";
        return categories.at(index);
    }
};

struct InputStats {  // include/input_stats.hpp:15-80
    uint32_t num_files = 0;
    std::map<uint8_t, uint64_t> records_per_bin, hashes_per_bin;
};

// Index container (include/index.hpp:19-138).  Keeps BOTH a plain interleaved copy and the
// Elias-Fano form; `use_ef` selects which one bulk_contains() reads.
struct Index {
    uint8_t window_size = 41, kmer_size = 19;
    double max_fpr = 0.01;
    InputSummary summary;
    InputStats stats;
    uint64_t bins = 0, technical_bins = 0, bin_size = 0, hash_shift = 0, bin_words = 0, hash_funs = 3;
    std::vector<uint64_t> plain;  // word (row*bin_words + b)
    SdVector ef;
    bool has_ef = false, has_plain = false;
    bool use_ef = false;

    void init_ibf(uint64_t nbins, uint64_t bsize, uint64_t nhash) {
        bins = nbins;
        bin_words = (nbins + 63) >> 6;
        technical_bins = bin_words << 6;
        bin_size = bsize;
        hash_shift = clz64(bsize);
        hash_funs = nhash;
        plain.assign(bin_size * bin_words, 0);
        has_plain = true;
    }
    void emplace(uint64_t value, uint64_t bin) {
        for (unsigned i = 0; i < hash_funs; ++i) {
            uint64_t row = hash_and_fit_row(value, IBF_SEEDS[i], bin_size, (unsigned)hash_shift);
            plain[row * bin_words + (bin >> 6)] |= 1ULL << (bin & 63);
        }
    }
    void compress() {  // uncompressed -> compressed conversion of include/index.hpp:43-50
        ef.build_from_words(plain, technical_bins * bin_size);
        has_ef = true;
    }
    void decompress() {
        plain.assign(bin_size * bin_words, 0);
        ef.for_each_one([&](uint64_t p) { plain[p >> 6] |= 1ULL << (p & 63); });
        has_plain = true;
    }
    // membership_agent::bulk_contains: out[b] = AND_i word(row_i, b)
    void bulk_contains(uint64_t value, uint64_t *out) const {
        for (uint64_t b = 0; b < bin_words; ++b) out[b] = ~0ULL;
        for (unsigned i = 0; i < hash_funs; ++i) {
            uint64_t row = hash_and_fit_row(value, IBF_SEEDS[i], bin_size, (unsigned)hash_shift);
            for (uint64_t b = 0; b < bin_words; ++b)
                out[b] &= use_ef ? ef.get_int(row * technical_bins + 64 * b) : plain[row * bin_words + b];
        }
    }
    uint8_t get_host_index() const {  // include/index.hpp:72-80
        return std::min(summary.category_index("host"), summary.category_index("human"));
    }
};

// src/utils.cpp:75-90
inline uint64_t bin_size_in_bits(uint64_t num_elements, unsigned num_hash, double max_fpr,
                                 uint64_t bits = 4294967293ULL) {
    double numerator = -(double)(num_elements * num_hash);
    double denominator = std::log(1 - std::exp(std::log(max_fpr) / num_hash));
    double result = std::ceil(numerator / denominator);
    if (result > (double)bits) return bits;
    return (uint64_t)result;
}

// ---------------------------------------------------------------------------------------------
// cereal BinaryArchive restatement -- little-endian raw fields.
// ---------------------------------------------------------------------------------------------
struct BinWriter {
    std::ostream &os;
    explicit BinWriter(std::ostream &o) : os(o) {}
    template <class T> void pod(const T &v) { os.write(reinterpret_cast<const char *>(&v), sizeof(T)); }
    void str(const std::string &s) { pod<uint64_t>(s.size()); os.write(s.data(), (std::streamsize)s.size()); }
    void int_vector(uint8_t width, uint64_t bit_size, const std::vector<uint64_t> &words) {
        uint64_t n_words = (bit_size + 63) >> 6;
        pod<uint8_t>(width);
        pod<float>(1.5f);
        pod<uint64_t>(n_words);
        pod<uint64_t>(bit_size);
        os.write(reinterpret_cast<const char *>(words.data()), (std::streamsize)(n_words * 8));
    }
    void int_vector(const IntVec &v) { int_vector(v.width, v.bit_size(), v.words); }
    // select_support_mcl's cereal save: arg_cnt, logn, logn2, logn4; then, if there are args, m_superblock, mini_or_long
    // (bit i = superblock i has a miniblock; EMPTY when no long superblock exists) and per superblock its long or mini vector
    void select_mcl(const SelectMcl &s) {
        pod<uint64_t>(s.m_arg_cnt);
        pod<uint32_t>(s.m_logn); pod<uint32_t>(s.m_logn2); pod<uint32_t>(s.m_logn4);
        const uint64_t sb = (s.m_arg_cnt + 4095) >> 12;
        if (!s.m_arg_cnt) return;
        int_vector(s.m_superblock);
        IntVec mini_or_long(0, 0, 1);
        if (!s.m_longsuperblock.empty()) {
            mini_or_long = IntVec(sb, 0, 1);
            for (uint64_t i = 0; i < sb; ++i) mini_or_long.set(i, !s.m_miniblock[i].empty());
        }
        int_vector(mini_or_long);
        for (uint64_t i = 0; i < sb; ++i) {
            if (!mini_or_long.empty() && !mini_or_long.get(i)) int_vector(s.m_longsuperblock[i]);
            else int_vector(s.m_miniblock[i]);
        }
    }
};
struct BinReader {
    std::istream &is;
    explicit BinReader(std::istream &i) : is(i) {}
    template <class T> T pod() {
        T v;
        is.read(reinterpret_cast<char *>(&v), sizeof(T));
        if (!is) throw std::runtime_error("index file truncated at offset " + std::to_string((long long)is.tellg()));
        return v;
    }
    std::string str() {
        uint64_t n = pod<uint64_t>();
        if (n > (1u << 20)) throw std::runtime_error("implausible string length in index file");
        std::string s(n, '\0');
        is.read(&s[0], (std::streamsize)n);
        return s;
    }
    void int_vector(uint8_t &width, uint64_t &bit_size, std::vector<uint64_t> &words) {
        width = pod<uint8_t>();
        float gf = pod<float>();
        uint64_t n_words = pod<uint64_t>();
        bit_size = pod<uint64_t>();
        if (width < 1 || width > 64 || gf != 1.5f || n_words * 64 < bit_size)
            throw std::runtime_error("int_vector framing check failed near offset " + std::to_string((long long)is.tellg()));
        words.assign(n_words + 1, 0);
        is.read(reinterpret_cast<char *>(words.data()), (std::streamsize)(n_words * 8));
        if (!is) throw std::runtime_error("index file truncated inside int_vector");
    }
    IntVec int_vec() {
        IntVec v; uint64_t bits;
        int_vector(v.width, bits, v.words);
        if (bits % v.width) throw std::runtime_error("int_vector bit size is not a multiple of its width");
        v.n = bits / v.width;
        return v;
    }
    bool at_end() { return is.peek() == std::char_traits<char>::eof(); }
    // select_support_mcl's cereal load
    void select_mcl(SelectMcl &s) {
        s.m_arg_cnt = pod<uint64_t>();
        s.m_logn = pod<uint32_t>(); s.m_logn2 = pod<uint32_t>(); s.m_logn4 = pod<uint32_t>();
        const uint64_t sb = (s.m_arg_cnt + 4095) >> 12;
        s.m_longsuperblock.clear(); s.m_miniblock.clear();
        if (!s.m_arg_cnt) return;
        s.m_superblock = int_vec();
        if (s.m_superblock.n != sb) throw std::runtime_error("select_support_mcl: m_superblock size");
        const IntVec mini_or_long = int_vec();
        if (mini_or_long.width != 1 || (mini_or_long.n != 0 && mini_or_long.n != sb)) throw std::runtime_error("select_support_mcl: mini_or_long size");
        if (!mini_or_long.empty()) s.m_longsuperblock.assign(sb, IntVec());
        s.m_miniblock.assign(sb, IntVec());
        for (uint64_t i = 0; i < sb; ++i) {
            if (!mini_or_long.empty() && !mini_or_long.get(i)) s.m_longsuperblock[i] = int_vec();
            else s.m_miniblock[i] = int_vec();
        }
    }
};

// include/store_index.hpp:12-17 + Index::serialize include/index.hpp:122-138
inline void store_index(const std::string &path, const Index &idx) {
    if (!idx.has_ef) throw std::runtime_error("store_index: compress() first");
    std::ofstream os(path, std::ios::binary);
    BinWriter w(os);
    w.pod<uint8_t>(idx.window_size);
    w.pod<uint8_t>(idx.kmer_size);
    w.pod<double>(idx.max_fpr);
    w.pod<uint8_t>(idx.summary.num_bins);
    w.pod<uint64_t>(idx.summary.categories.size());
    for (auto &c : idx.summary.categories) w.str(c);
    w.pod<uint64_t>(idx.summary.filepath_to_bin.size());
    for (auto &p : idx.summary.filepath_to_bin) { w.str(p.first); w.pod<uint8_t>(p.second); }
    w.pod<uint64_t>(idx.summary.bin_to_category.size());
    for (auto &p : idx.summary.bin_to_category) { w.pod<uint8_t>(p.first); w.str(p.second); }
    w.pod<uint32_t>(idx.stats.num_files);
    w.pod<uint64_t>(idx.stats.records_per_bin.size());
    for (auto &p : idx.stats.records_per_bin) { w.pod<uint8_t>(p.first); w.pod<uint64_t>(p.second); }
    w.pod<uint64_t>(idx.stats.hashes_per_bin.size());
    for (auto &p : idx.stats.hashes_per_bin) { w.pod<uint8_t>(p.first); w.pod<uint64_t>(p.second); }
    w.pod<uint64_t>(idx.bins);
    w.pod<uint64_t>(idx.technical_bins);
    w.pod<uint64_t>(idx.bin_size);
    w.pod<uint64_t>(idx.hash_shift);
    w.pod<uint64_t>(idx.bin_words);
    w.pod<uint64_t>(idx.hash_funs);
    w.pod<uint64_t>(idx.ef.size);
    w.pod<uint8_t>(idx.ef.wl);
    w.int_vector(idx.ef.wl, idx.ef.ones * idx.ef.wl, idx.ef.low);
    w.int_vector(1, idx.ef.high_bits, idx.ef.high);
    // m_high_1_select, m_high_0_select: the sd_vector's two select_support_mcl structures over m_high
    for (int b = 1; b >= 0; --b) {
        SelectMcl sel;
        sel.init(idx.ef.high.data(), idx.ef.high_bits, b);
        w.select_mcl(sel);
    }
}

// src/load_index.cpp:8-15
inline void load_index(Index &idx, const std::string &path) {
    std::ifstream is(path, std::ios::binary);
    if (!is) throw std::runtime_error("cannot open index " + path);
    BinReader r(is);
    idx.window_size = r.pod<uint8_t>();
    idx.kmer_size = r.pod<uint8_t>();
    idx.max_fpr = r.pod<double>();
    idx.summary = InputSummary();
    idx.summary.num_bins = r.pod<uint8_t>();
    uint64_t n = r.pod<uint64_t>();
    for (uint64_t i = 0; i < n; ++i) idx.summary.categories.push_back(r.str());
    n = r.pod<uint64_t>();
    for (uint64_t i = 0; i < n; ++i) { std::string s = r.str(); uint8_t b = r.pod<uint8_t>(); idx.summary.filepath_to_bin.emplace_back(s, b); }
    n = r.pod<uint64_t>();
    for (uint64_t i = 0; i < n; ++i) { uint8_t b = r.pod<uint8_t>(); idx.summary.bin_to_category[b] = r.str(); }
    idx.stats = InputStats();
    idx.stats.num_files = r.pod<uint32_t>();
    n = r.pod<uint64_t>();
    for (uint64_t i = 0; i < n; ++i) { uint8_t b = r.pod<uint8_t>(); idx.stats.records_per_bin[b] = r.pod<uint64_t>(); }
    n = r.pod<uint64_t>();
    for (uint64_t i = 0; i < n; ++i) { uint8_t b = r.pod<uint8_t>(); idx.stats.hashes_per_bin[b] = r.pod<uint64_t>(); }
    idx.bins = r.pod<uint64_t>();
    idx.technical_bins = r.pod<uint64_t>();
    idx.bin_size = r.pod<uint64_t>();
    idx.hash_shift = r.pod<uint64_t>();
    idx.bin_words = r.pod<uint64_t>();
    idx.hash_funs = r.pod<uint64_t>();
    if (idx.technical_bins != ((idx.bins + 63) >> 6) << 6 || idx.bin_words != idx.technical_bins >> 6 ||
        idx.hash_shift != clz64(idx.bin_size) || idx.hash_funs < 1 || idx.hash_funs > 5)
        throw std::runtime_error("IBF header self-check failed");
    idx.ef = SdVector();
    idx.ef.size = r.pod<uint64_t>();
    idx.ef.wl = r.pod<uint8_t>();
    if (idx.ef.size != idx.technical_bins * idx.bin_size) throw std::runtime_error("sd_vector size != TB*S");
    uint8_t width; uint64_t bits;
    r.int_vector(width, bits, idx.ef.low);
    if (width != idx.ef.wl && !(idx.ef.wl == 0)) throw std::runtime_error("m_low width != m_wl");
    idx.ef.ones = idx.ef.wl ? bits / idx.ef.wl : 0;
    r.int_vector(width, bits, idx.ef.high);
    if (width != 1) throw std::runtime_error("m_high is not a bit vector");
    idx.ef.high_bits = bits;
    // m_high_1_select / m_high_0_select, when the file carries them (files of an earlier build of this project end behind m_high)
    idx.ef.has_sel = false;
    if (!r.at_end()) {
        r.select_mcl(idx.ef.sel1);
        r.select_mcl(idx.ef.sel0);
        if (!r.at_end()) throw std::runtime_error("bytes follow m_high_0_select");
        idx.ef.sel1.v_words = idx.ef.sel0.v_words = idx.ef.high.data();
        idx.ef.sel1.v_size = idx.ef.sel0.v_size = idx.ef.high_bits;
        idx.ef.sel1.t_b = 1; idx.ef.sel0.t_b = 0;
        idx.ef.has_sel = true;
    }
    idx.ef.build_select();
    idx.has_ef = true;
    idx.decompress();
}

// ---------------------------------------------------------------------------------------------
// Arguments (include/dehost_arguments.hpp:9-43)
// ---------------------------------------------------------------------------------------------
struct DehostArguments {
    std::string read_file, read_file2, db, category_to_extract, prefix, log_file = "charon.log", dist = "kde";
    bool is_paired = false, run_extract = false;
    uint8_t chunk_size = 100;
    float lo_hi_threshold = 0.15f;
    uint16_t num_reads_to_fit = 5000;
    float min_quality = 15.0f;
    uint32_t min_length = 140;
    float min_compression = 0;
    uint8_t confidence_threshold = 7;
    float confidence_probability_threshold = 0;
    float host_unique_prop_lo_threshold = 0.05f;
    float min_proportion_difference = 0.04f;
    float min_prob_difference = 0;
    uint8_t threads = 1, verbosity = 0;
    uint8_t min_hits = 0;  // StatsModel::min_hits_ is never initialised in the reference (UB); 0 here
    bool skip_gzip = false;  // oracle-only switch: leave `compression` at 0 (hot-path-only timing)
    // `charon classify` (src/classify_main.cpp, include/classify_arguments.hpp): the same loop and Result state machine, every read
    // goes through ReadEntry::classify (call_category); only the defaults differ (see classify_defaults below)
    bool classify_mode = false;
};
inline DehostArguments classify_defaults() {  // include/classify_arguments.hpp:19-29 mapped onto the shared argument struct
    DehostArguments o;
    o.classify_mode = true; o.dist = "beta"; o.min_quality = 10.0f; o.min_length = 140; o.min_compression = 0.15f;
    o.confidence_threshold = 2; o.min_proportion_difference = 0.0f;
    // the remaining StatsModel members are not set by the ClassifyArguments constructor (include/classify_stats.hpp:428-440)
    // and call_category does not read them
    return o;
}

// default KDE training tables: src/dehost_main.cpp:23-206 (sorted by the KDEParams constructor,
// include/classify_stats.hpp:214-218).  Loaded from tests/golden/default_kde.txt-style data at run time.
struct DefaultTables {
    std::vector<float> pos, neg;
};
inline DefaultTables &default_tables() { static DefaultTables t; return t; }
inline void load_default_tables(const std::string &path) {
    std::ifstream is(path);
    if (!is) throw std::runtime_error("cannot open KDE table file " + path);
    DefaultTables &t = default_tables();
    t.pos.clear(); t.neg.clear();
    std::string tag; size_t n;
    while (is >> tag >> n) {
        std::vector<float> &dst = (tag == "pos") ? t.pos : t.neg;
        for (size_t i = 0; i < n; ++i) { std::string tok; is >> tok; dst.push_back((float)std::strtod(tok.c_str(), nullptr)); }  // double literal narrowed to float
    }
}

// stats::dexp(x, 300) in float (include/classify_stats.hpp:371).  The
// exp(log-density) form of statslib 3.x is used; the alternative differs by a few float ulps.
inline float dexp300(float x) {
    if (std::isnan(x)) return std::numeric_limits<float>::quiet_NaN();
    if (x < 0.0f) return 0.0f;
    return std::exp(std::log(300.0f) - 300.0f * x);
}

struct KDEParams {  // include/classify_stats.hpp:210-254
    std::vector<float> dataset;
    float h;
    KDEParams(const std::vector<float> &d, float h_) : dataset(d), h(h_) { std::sort(dataset.begin(), dataset.end()); }
    void fit(const std::vector<float> &training) { dataset = training; }  // :234-240 (h unchanged, NOT re-sorted)
    float K(const float &x) const { return (float)(std::exp(-std::pow((double)x, 2) / 2) / std::sqrt(2 * 3.141592653589793238463)); }
    float prob(const float &x) const {
        float total_sum = 0;
        for (const float &xi : dataset) total_sum += K((x - xi) / h);
        return total_sum / (h * dataset.size());
    }
};

struct ProbPair { double pos, neg; };

struct TrainingData {  // include/classify_stats.hpp:34-114
    bool complete = false, pos_complete = false, neg_complete = false;
    uint16_t num_reads_to_fit = 5000;
    std::vector<float> pos, neg;
    bool check_status() {
        if (pos.size() >= num_reads_to_fit) pos_complete = true;
        if (neg.size() >= num_reads_to_fit) neg_complete = true;
        if (pos_complete && neg_complete) complete = true;
        return complete;
    }
    bool add_pos(float val) { if (pos.size() < num_reads_to_fit) pos.push_back(val); else check_status(); return complete; }
    bool add_neg(float val) { if (neg.size() < num_reads_to_fit && val > 0) neg.push_back(val); else check_status(); return complete; }
    void clear() { pos.clear(); neg.clear(); }
};

// mean / variance helpers of include/classify_stats.hpp:20-32 (double sum of float data; the variance lambda's accumulator is a float)
inline double mean_of(const std::vector<float> &v) {
    double sum = 0.0;
    for (float x : v) sum += x;
    return v.empty() ? 0 : sum / v.size();
}
inline double variance_of(const std::vector<float> &v, double mean) {
    double sum = 0.0;
    for (float val : v) { const float accumulator = (float)sum; sum = accumulator + ((val - mean) * (val - mean)); }
    return v.size() <= 1 ? 0 : sum / (v.size() - 1);
}
// stats::dgamma(x, shape, scale) / stats::dbeta(x, a, b) of kthohr/stats 3.4.0 called with float arguments
// (include/classify_stats.hpp:377-381) [3P-recall: statslib is absent from the image -- parity unpinned]: sanity checks -> NaN,
// the boundary cases of the support, otherwise exp(log-density), evaluated in float.  tests/test_oracle_kat.py cross-checks the
// densities against scipy.stats.
inline float stats_dgamma(float x, float shape, float scale) {
    if (std::isnan(x) || std::isnan(shape) || std::isnan(scale) || shape < 0.0f || scale < 0.0f) return std::numeric_limits<float>::quiet_NaN();
    if (x < 0.0f) return 0.0f;
    if (x == 0.0f) return shape < 1.0f ? std::numeric_limits<float>::infinity() : (shape > 1.0f ? 0.0f : 1.0f / scale);
    if (std::isinf(x)) return 0.0f;
    return std::exp(-std::lgamma(shape) - shape * std::log(scale) + (shape - 1.0f) * std::log(x) - x / scale);
}
inline float stats_dbeta(float x, float a, float b) {
    const float inf = std::numeric_limits<float>::infinity();
    if (std::isnan(x) || std::isnan(a) || std::isnan(b) || a < 0.0f || b < 0.0f) return std::numeric_limits<float>::quiet_NaN();
    if (x < 0.0f || x > 1.0f) return 0.0f;
    if (a == 0.0f && b == 0.0f) return (x == 0.0f || x == 1.0f) ? inf : 0.0f;
    if (a == 0.0f || (std::isinf(b) && !std::isinf(a))) return x == 0.0f ? inf : 0.0f;
    if (b == 0.0f || (std::isinf(a) && !std::isinf(b))) return x == 1.0f ? inf : 0.0f;
    if (std::isinf(a) && std::isinf(b)) return x == 0.5f ? inf : 0.0f;
    if (x == 0.0f) return a < 1.0f ? inf : (a > 1.0f ? 0.0f : b);
    if (x == 1.0f) return b < 1.0f ? inf : (b > 1.0f ? 0.0f : a);
    return std::exp(-(std::lgamma(a) + std::lgamma(b) - std::lgamma(a + b)) + (a - 1.0f) * std::log(x) + (b - 1.0f) * std::log(1.0f - x));
}
struct GammaParams {  // include/classify_stats.hpp:116-157
    float shape, loc, scale;
    void fit(const std::vector<float> &data) {  // :127-137
        const double mu = mean_of(data), ln_mu = std::log(mu);
        double sum_ln = 0.0;
        for (float n : data) sum_ln += std::log(n);  // std::log(float) -> float, accumulated in double
        const double mean_ln = data.empty() ? 0 : sum_ln / data.size();
        const double s = ln_mu - mean_ln;
        shape = (float)((3 - s + std::sqrt((s - 3) * (s - 3) + 24 * s)) / (12 * s));
        scale = (float)(mu / shape);
    }
    void fit_loc(const std::vector<float> &data) { loc = (float)(mean_of(data) - (shape * scale)); }  // :139-142
};
struct BetaParams {  // :159-208
    float alpha, beta, loc;
    void fit(const std::vector<float> &data) {  // :171-191 (the asserts vanish in a release build)
        const double mu = mean_of(data), var = variance_of(data, mu);
        alpha = (float)(mu * ((mu * (1 - mu) / var) - 1));
        beta = (float)((1 - mu) * ((mu * (1 - mu) / var) - 1));
    }
};

struct Model {  // include/classify_stats.hpp:261-393
    bool ready = false;
    std::string dist = "kde";
    GammaParams g_pos{25, 0, 0.02f}, g_neg{10, 0, 0.005f};  // :265-266
    BetaParams b_pos{6, 4, 0}, b_neg{6, 40, 0};              // :267-268
    KDEParams k_pos, k_neg;
    Model() : k_pos(default_tables().pos, 0.1f), k_neg(default_tables().neg, 0.001f) {}
    explicit Model(const std::string &d) : dist(d), k_pos(default_tables().pos, 0.1f), k_neg(default_tables().neg, 0.001f) {}
    void train(TrainingData &td) {  // :289-368
        if (dist == "kde") {
            if (td.pos_complete) k_pos.fit(td.pos);
            if (td.neg_complete) k_neg.fit(td.neg);
        } else if (dist == "beta") {
            if (td.pos_complete) b_pos.fit(td.pos);
            if (td.neg_complete) b_neg.fit(td.neg);
        } else {  // gamma: an incomplete neg set still moves the default distribution's location (:306-312)
            if (td.pos_complete) g_pos.fit(td.pos);
            if (td.neg_complete) g_neg.fit(td.neg); else g_neg.fit_loc(td.neg);
        }
        ready = true;
        td.clear();
    }
    ProbPair prob(const float &x) const {  // :370-389
        const float p_err = dexp300(x);
        float p_pos, p_neg;
        if (dist == "kde") { p_pos = k_pos.prob(x); p_neg = k_neg.prob(x); }
        else if (dist == "gamma") { p_pos = stats_dgamma(x - g_pos.loc, g_pos.shape, g_pos.scale); p_neg = stats_dgamma(x - g_neg.loc, g_neg.shape, g_neg.scale); }
        else { p_pos = stats_dbeta(x, b_pos.alpha, b_pos.beta); p_neg = stats_dbeta(x, b_neg.alpha, b_neg.beta); }
        if (x == 1) p_pos = 1;
        const float total = p_err + p_pos + p_neg;
        ProbPair pp; pp.pos = p_pos / total; pp.neg = (p_err + p_neg) / total;
        return pp;
    }
};

struct StatsModel {  // include/classify_stats.hpp:395-584
    bool ready_ = false;
    float lo_hi_threshold_, min_quality_;
    uint32_t min_length_;
    float min_compression_;
    int8_t confidence_threshold_;  // narrowed from uint8_t (:404,497)
    float confidence_probability_threshold_;
    uint8_t min_hits_;
    float host_unique_prop_lo_threshold_, min_proportion_difference_, min_prob_difference_;
    std::vector<TrainingData> training_data_;
    std::vector<Model> models_;

    StatsModel() {}
    StatsModel(const DehostArguments &opt, const InputSummary &summary)
        : lo_hi_threshold_(opt.lo_hi_threshold), min_quality_(opt.min_quality), min_length_(opt.min_length),
          min_compression_(opt.min_compression), confidence_threshold_((int8_t)opt.confidence_threshold),
          confidence_probability_threshold_(opt.confidence_probability_threshold), min_hits_(opt.min_hits),
          host_unique_prop_lo_threshold_(opt.host_unique_prop_lo_threshold),
          min_proportion_difference_(opt.min_proportion_difference), min_prob_difference_(opt.min_prob_difference) {
        for (unsigned i = 0; i < summary.num_categories(); ++i) {
            models_.emplace_back(opt.dist);
            TrainingData td; td.num_reads_to_fit = opt.num_reads_to_fit;
            training_data_.push_back(td);
        }
    }
    void force_ready() {  // :463-473
        for (size_t i = 0; i < models_.size(); ++i) if (!models_[i].ready) models_[i].train(training_data_[i]);
        ready_ = true;
    }
    void check_if_ready() { if (ready_) return; for (auto &m : models_) if (!m.ready) return; ready_ = true; }
    void train_model_at(uint8_t i) { models_[i].train(training_data_[i]); check_if_ready(); }  // :521-533
    bool add_read_to_training_data(const std::vector<float> &props) {  // :535-578
        uint8_t pos_i = 255;
        double max_val = 0.0;
        int num_above = 0;
        for (uint8_t i = 0; i < props.size(); ++i) {
            const float &val = props.at(i);
            if (val > lo_hi_threshold_) num_above += 1;
            if (val == max_val) pos_i = 255;
            else if (val > max_val) { pos_i = i; max_val = val; }
        }
        bool add_pos = (pos_i != 255 && num_above == 1);
        bool add_neg = add_pos || (num_above == 0);
        if (add_pos) {
            bool rtt = training_data_.at(pos_i).add_pos(props[pos_i]);
            if (rtt && !models_[pos_i].ready) train_model_at(pos_i);
        }
        if (add_neg) {
            for (uint8_t i = 0; i < props.size(); ++i) {
                if (i != pos_i) {
                    bool rtt = training_data_.at(i).add_neg(props[i]);
                    if (rtt && !models_[i].ready) train_model_at(i);
                }
            }
        }
        return ready_;
    }
    ProbPair classify(size_t i, const float &x) const { return models_.at(i).prob(x); }
};

// src/utils.cpp:114-124 via gzip-hpp: deflateInit2(Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15+16, 8,
// Z_DEFAULT_STRATEGY), one deflate(Z_FINISH).
inline float get_compression_ratio(const std::string &sequence) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK)
        throw std::runtime_error("deflateInit2 failed");
    std::vector<unsigned char> out(deflateBound(&zs, (uLong)sequence.size()) + 64);
    zs.next_in = (Bytef *)sequence.data();
    zs.avail_in = (uInt)sequence.size();
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    size_t compressed = out.size() - zs.avail_out;
    deflateEnd(&zs);
    return (float)((double)compressed / (double)sequence.size());
}

// ---------------------------------------------------------------------------------------------
// ReadEntry (include/read_entry.hpp)
// ---------------------------------------------------------------------------------------------
struct ReadEntry {
    std::string read_id_;
    uint32_t length_ = 0;
    float mean_quality_ = 0, compression_ = 0;
    uint32_t num_hashes_ = 0;
    std::vector<std::vector<uint64_t>> bits_;  // one bit-row (bin_words words) per minimiser, copied (:86-90)
    std::vector<uint32_t> counts_, unique_counts_;
    std::vector<float> proportions_, unique_proportions_;
    std::vector<double> probabilities_;
    uint8_t call_ = 255, confidence_score_ = 0;

    ReadEntry() {}
    ReadEntry(const std::string &id, uint32_t length, float mq, float comp, const InputSummary &s)  // :46-64
        : read_id_(id), length_(length), mean_quality_(mq), compression_(comp), counts_(s.num_categories(), 0),
          unique_counts_(s.num_categories(), 0), proportions_(s.num_categories(), 0),
          unique_proportions_(s.num_categories(), 0), probabilities_(s.num_categories(), 1) {
        bits_.reserve(length);
    }
    void update_entry(const uint64_t *row, size_t words) { bits_.emplace_back(row, row + words); num_hashes_ += 1; }

    void get_counts(const InputSummary &summary) {  // :92-138
        std::vector<uint64_t> total(summary.num_bins, 0);
        for (const auto &entry : bits_)
            for (unsigned b = 0; b < summary.num_bins; ++b) total[b] += (entry[b >> 6] >> (b & 63)) & 1;
        std::vector<uint8_t> index_per_category(summary.num_categories(), 255);
        for (unsigned bin = 0; bin < total.size(); ++bin) {
            const std::string &category = summary.bin_to_category.at((uint8_t)bin);
            uint8_t index = summary.category_index(category);
            if (index_per_category.at(index) == 255 || total[bin] > total[index_per_category[index]]) {
                index_per_category.at(index) = (uint8_t)bin;
                counts_.at(index) = (uint32_t)total[bin];
            }
        }
        std::vector<uint8_t> found;
        for (const auto &entry : bits_) {
            found.clear();
            for (unsigned c = 0; c < index_per_category.size(); ++c) {
                unsigned ci = index_per_category[c];
                if (ci != 255 && ((entry[ci >> 6] >> (ci & 63)) & 1)) found.push_back((uint8_t)c);
            }
            if (found.size() == 1) unique_counts_.at(found.front()) += 1;
        }
    }
    void get_proportions() {  // :140-150
        for (size_t i = 0; i < proportions_.size(); ++i) {
            proportions_[i] = static_cast<float>(counts_[i]) / static_cast<float>(num_hashes_);
            unique_proportions_[i] = static_cast<float>(unique_counts_[i]) / static_cast<float>(num_hashes_);
        }
    }
    void post_process(const InputSummary &s) { get_counts(s); get_proportions(); }

    void apply_model(const StatsModel &m) {  // :271-279
        for (size_t i = 0; i < unique_proportions_.size(); ++i) probabilities_[i] *= m.classify(i, unique_proportions_[i]).pos;
    }
    void call_category(const StatsModel &m) {  // :157-216
        uint8_t first = 0, second = 1;
        if (unique_counts_.at(second) > unique_counts_.at(first)) std::swap(first, second);
        for (size_t i = 2; i < unique_counts_.size(); ++i) {
            if (unique_counts_[i] > unique_counts_[second]) {
                second = (uint8_t)i;
                if (unique_counts_[second] > unique_counts_[first]) std::swap(first, second);
            }
        }
        uint32_t raw = unique_counts_[first] - unique_counts_[second];
        confidence_score_ = raw > 255 ? 255 : (uint8_t)raw;
        if (mean_quality_ < m.min_quality_) return;
        if (length_ < m.min_length_) return;
        if (compression_ < m.min_compression_) return;
        if (probabilities_[second] == 0 && probabilities_[first] > 0) call_ = first;
        else if (confidence_score_ > m.confidence_threshold_ && probabilities_[first] > probabilities_[second]) call_ = first;
        const uint32_t fc = counts_[first], sc = counts_[second];
        if (sc > fc || fc - sc < m.min_hits_) call_ = 255;
        const float fp = proportions_[first], sp = proportions_[second];
        if (sp > fp || fp - sp < m.min_proportion_difference_) call_ = 255;
    }
    void call_host(const StatsModel &m, uint8_t host_index) {  // :218-269
        const uint8_t other_index = 1 - host_index;
        double hu = unique_proportions_.at(host_index), ou = unique_proportions_.at(other_index);
        double hp = probabilities_.at(host_index), op = probabilities_.at(other_index);
        uint8_t first = host_index, second = other_index;
        if (hu < ou) { first = other_index; second = host_index; }
        uint32_t raw = unique_counts_.at(first) - unique_counts_.at(second);  // unsigned wrap reproduced
        confidence_score_ = raw > 255 ? 255 : (uint8_t)raw;
        if (confidence_score_ < m.confidence_threshold_) return;  // int promotion: uint8 vs int8
        if (mean_quality_ < m.min_quality_) return;
        if (length_ < m.min_length_) return;
        if (compression_ < m.min_compression_) return;
        if (hu > ou && hu - ou > m.min_proportion_difference_ && hp > op && hp - op > m.min_prob_difference_ &&
            std::max(hp * confidence_score_, static_cast<double>(confidence_score_)) >= m.confidence_probability_threshold_)
            call_ = host_index;
        else if (hu < m.host_unique_prop_lo_threshold_ && hu < ou && ou - hu > m.min_proportion_difference_ && hp < op &&
                 op - hp > m.min_prob_difference_ &&
                 std::max(op * confidence_score_, static_cast<double>(confidence_score_)) >= m.confidence_probability_threshold_)
            call_ = other_index;
    }
    void classify(const StatsModel &m) { apply_model(m); call_category(m); }
    void dehost(const StatsModel &m, uint8_t host_index) { apply_model(m); call_host(m, host_index); }

    void print_assignment_result(const InputSummary &s, std::ostream &os) const {  // :322-337
        os << (call_ == 255 ? "U" : "C") << "\t";
        os.precision(6);
        os << read_id_ << "\t" << s.category_name(call_) << "\t" << length_ << "\t" << num_hashes_ << "\t" << mean_quality_
           << "\t" << +confidence_score_ << "\t" << compression_ << "\t";
        for (unsigned i = 0; i < s.num_categories(); i++)
            os << s.categories.at(i) << ":" << counts_.at(i) << ":" << proportions_.at(i) << ":" << unique_proportions_.at(i)
               << ":" << probabilities_.at(i) << " ";
        os << "\n";
    }
};

// ---------------------------------------------------------------------------------------------
// Result state machine (include/result.hpp).  Extract writers are out of scope; only the
// capacity effect of --extract on the training cache (:80-85) is reproduced.
// ---------------------------------------------------------------------------------------------
struct Result {
    InputSummary summary_;
    StatsModel stats_model_;
    std::vector<ReadEntry> cached_;
    size_t cache_capacity_ = 0;
    std::vector<uint64_t> classified_counts;
    uint64_t unclassified_count = 0;
    std::ostream *os;
    std::vector<ReadEntry> *sink = nullptr;  // optional: collect classified entries instead of/in addition to printing

    Result(const DehostArguments &opt, const InputSummary &s, std::ostream &o)
        : summary_(s), stats_model_(opt, s), classified_counts(s.num_categories(), 0), os(&o) {
        if (opt.run_extract) cache_capacity_ = (size_t)opt.num_reads_to_fit * s.num_categories() * 4;
    }
    uint8_t classify_read(ReadEntry &e, bool dehost) {  // :97-116
        if (dehost) e.dehost(stats_model_, summary_.host_category_index());
        else e.classify(stats_model_);
        if (os) e.print_assignment_result(summary_, *os);
        if (sink) sink->push_back(e);
        if (e.call_ < 255) classified_counts[e.call_] += 1; else unclassified_count += 1;
        return e.call_;
    }
    void classify_cache(bool dehost) {  // :181-198
        for (auto e : cached_) classify_read(e, dehost);
        cached_.resize(0);
    }
    void add_read(ReadEntry &e, bool dehost) {  // :130-153 (paired :155-179 is identical but for the records)
        if (stats_model_.ready_) { classify_read(e, dehost); return; }
        bool training_complete = false;
        if (cached_.size() < cache_capacity_) {
            cached_.push_back(e);
            training_complete = stats_model_.add_read_to_training_data(e.unique_proportions_);
        } else {
            stats_model_.force_ready();  // the read that arrives here is dropped (quirk A.9)
            training_complete = true;
        }
        if (training_complete) classify_cache(dehost);
    }
    void complete(bool dehost) { classify_cache(dehost); }  // :200-202
};

// ---------------------------------------------------------------------------------------------
// FASTA/FASTQ records (seqan3::sequence_file_input<my_traits> semantics, whole-file reader; gz via zlib)
// ---------------------------------------------------------------------------------------------
struct Record { std::string id, seq, qual; };

inline std::string slurp_maybe_gz(const std::string &path) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string data;
    char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof buf)) > 0) data.append(buf, (size_t)n);
    gzclose(f);
    return data;
}
inline bool is_fastq_name(std::string p) {
    auto ends = [&](const std::string &s) { return p.size() >= s.size() && p.compare(p.size() - s.size(), s.size(), s) == 0; };
    if (ends(".gz")) p.resize(p.size() - 3);
    if (ends(".bz2")) p.resize(p.size() - 4);
    return ends(".fastq") || ends(".fq");
}
inline std::vector<Record> read_fastx(const std::string &path) {
    std::string data = slurp_maybe_gz(path);
    bool fq = is_fastq_name(path);
    std::vector<Record> recs;
    size_t p = 0, n = data.size();
    auto getline = [&](std::string &line) -> bool {
        if (p >= n) return false;
        size_t e = data.find('\n', p);
        if (e == std::string::npos) e = n;
        line.assign(data, p, e - p);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        p = e + 1;
        return true;
    };
    std::string line;
    if (fq) {
        while (getline(line)) {
            if (line.empty()) continue;
            if (line[0] != '@') throw std::runtime_error("FASTQ parse error: expected '@'");
            Record r; r.id = line.substr(1);
            // sequence lines until '+'
            while (getline(line) && (line.empty() || line[0] != '+')) r.seq += line;
            while (r.qual.size() < r.seq.size() && getline(line)) r.qual += line;
            for (char c : r.seq) if (!dna5_char_valid(c)) throw std::runtime_error("parse error: illegal character in sequence");
            recs.push_back(std::move(r));
        }
    } else {
        Record cur; bool have = false;
        while (getline(line)) {
            if (!line.empty() && (line[0] == '>' || line[0] == ';')) {
                if (have) recs.push_back(std::move(cur));
                cur = Record(); cur.id = line.substr(1); have = true;
            } else if (have) {
                for (char c : line) {
                    if (c == ' ' || c == '\t' || (c >= '0' && c <= '9')) continue;  // seqan3 FASTA skips blanks/digits
                    if (!dna5_char_valid(c)) throw std::runtime_error("parse error: illegal character in sequence");
                    cur.seq.push_back(c);
                }
            }
        }
        if (have) recs.push_back(std::move(cur));
    }
    return recs;
}

inline std::string first_token(const std::string &id) {  // split(id, " ")[0], src/utils.cpp:9-20
    size_t e = id.find(' ');
    return e == std::string::npos ? id : id.substr(0, e);
}
inline std::string upper_dna5(const std::string &s) {  // sequence_to_string, src/utils.cpp:105-112
    std::string o(s.size(), 'N');
    for (size_t i = 0; i < s.size(); ++i) o[i] = dna5_char(dna5_rank(s[i]));
    return o;
}
inline float mean_quality_of(const std::string &q1, const std::string &q2 = std::string()) {  // src/dehost_main.cpp:355-360
    int sum = 0;
    for (char c : q1) sum += (int)c - 33;
    for (char c : q2) sum += (int)c - 33;
    size_t n = q1.size() + q2.size();
    return n ? static_cast<float>(sum) / static_cast<float>(n) : 0.0f;
}

// the body of the parallel loop, src/dehost_main.cpp:345-373 (single) / :420-468 (paired)
inline ReadEntry process_read(const Index &index, const InputSummary &summary, const Record &r1, const Record *r2,
                              bool skip_gzip) {
    const std::string read_id = first_token(r1.id);
    uint32_t length = (uint32_t)(r1.seq.size() + (r2 ? r2->seq.size() : 0));
    float mq = mean_quality_of(r1.qual, r2 ? r2->qual : std::string());
    float comp = 0;
    if (!skip_gzip) comp = get_compression_ratio(upper_dna5(r1.seq) + (r2 ? upper_dna5(r2->seq) : std::string()));
    ReadEntry e(read_id, length, mq, comp, summary);
    const Alphabet al = alphabet_dna5();
    std::vector<uint64_t> row(index.bin_words);
    for (uint64_t v : minimiser_hash(ranks_of(r1.seq), index.kmer_size, index.window_size, al)) {
        index.bulk_contains(v, row.data());
        e.update_entry(row.data(), row.size());
    }
    if (r2)
        for (uint64_t v : minimiser_hash(ranks_of(r2->seq), index.kmer_size, index.window_size, al)) {
            index.bulk_contains(v, row.data());
            e.update_entry(row.data(), row.size());
        }
    e.post_process(summary);
    return e;
}

// dehost_reads / dehost_paired_reads (src/dehost_main.cpp:314-477) at -t N (OpenMP over a chunk).
inline void dehost_run(DehostArguments opt, const Index &index, std::ostream &os, std::vector<ReadEntry> *sink = nullptr,
                       const std::vector<Record> *recs1_in = nullptr, const std::vector<Record> *recs2_in = nullptr) {
    std::vector<Record> own1, own2;
    if (!recs1_in) { own1 = read_fastx(opt.read_file); recs1_in = &own1; }
    if (opt.is_paired && !recs2_in) { own2 = read_fastx(opt.read_file2); recs2_in = &own2; }
    const std::vector<Record> &recs1 = *recs1_in;
    Result result(opt, index.summary, os);
    result.sink = sink;
    const size_t chunk = opt.chunk_size ? opt.chunk_size : 1;
    for (size_t base = 0; base < recs1.size(); base += chunk) {
        size_t n = std::min(chunk, recs1.size() - base);
        std::vector<ReadEntry> entries(n);
        std::vector<char> skip(n, 0);
#pragma omp parallel for num_threads(opt.threads) schedule(dynamic, 1)
        for (long i = 0; i < (long)n; ++i) {
            const Record &r1 = recs1[base + i];
            const Record *r2 = nullptr;
            if (opt.is_paired) {
                r2 = &(*recs2_in).at(base + i);
                std::string id1 = r1.id, id2 = r2->id;
                if (!id1.empty()) id1.erase(id1.size() - 1);
                if (!id2.empty()) id2.erase(id2.size() - 1);
                if (id1 != id2) { std::fprintf(stderr, "Your pairs don't match for read ids.\n"); std::abort(); }
            }
            if (r1.seq.size() + (r2 ? r2->seq.size() : 0) == 0) { skip[i] = 1; continue; }
            entries[i] = process_read(index, result.summary_, r1, r2, opt.skip_gzip);
        }
        // critical(add_read_to_results): serial, in input order (= the reference at -t 1)
        for (size_t i = 0; i < n; ++i) {
            if (skip[i]) continue;
            // the paired dehost path calls add_paired_read without dehost=true (:470); classify never passes it (src/classify_main.cpp:180)
            result.add_read(entries[i], /*dehost=*/!opt.is_paired && !opt.classify_mode);
        }
    }
    result.complete(!opt.is_paired && !opt.classify_mode);  // :379 complete(true) / :475 complete()
}

// minimal `charon index` restatement (src/index_main.cpp:75-305) to fabricate test indexes.
// `order` fixes the category order explicitly (the reference takes unordered_set iteration order, :79,110).
inline Index build_index(const std::vector<std::pair<std::string, std::string>> &file_category,  // (fasta path, category)
                         const std::vector<std::string> &category_order, unsigned w = 41, unsigned k = 19,
                         unsigned num_hash = 3, double max_fpr = 0.01, uint64_t force_bin_size = 0) {
    Index idx;
    idx.window_size = (uint8_t)w; idx.kmer_size = (uint8_t)k; idx.max_fpr = max_fpr;
    idx.summary.categories = category_order;
    std::vector<std::vector<uint64_t>> hashes;
    uint8_t bin = 0;
    uint64_t max_h = 0;
    const Alphabet al = alphabet_dna5();
    for (auto &fc : file_category) {
        idx.summary.bin_to_category[bin] = fc.second;
        idx.summary.filepath_to_bin.emplace_back(fc.first, bin);
        std::vector<uint64_t> set;
        uint64_t nrec = 0;
        for (auto &rec : read_fastx(fc.first)) {
            ++nrec;
            auto m = minimiser_hash(ranks_of(rec.seq), k, w, al);
            set.insert(set.end(), m.begin(), m.end());
        }
        std::sort(set.begin(), set.end());
        set.erase(std::unique(set.begin(), set.end()), set.end());
        idx.stats.num_files += 1;
        idx.stats.records_per_bin[bin] = nrec;
        idx.stats.hashes_per_bin[bin] = set.size();
        max_h = std::max<uint64_t>(max_h, set.size());
        hashes.push_back(std::move(set));
        ++bin;
    }
    idx.summary.num_bins = bin;
    uint64_t S = force_bin_size ? force_bin_size : bin_size_in_bits(max_h, num_hash, max_fpr);
    if (S == 0) throw std::runtime_error("The size of a bin must be > 0.");  // seqan3 interleaved_bloom_filter constructor [3P]
    idx.init_ibf(bin, S, num_hash);
    for (unsigned b = 0; b < hashes.size(); ++b)
        for (uint64_t v : hashes[b]) idx.emplace(v, b);
    idx.compress();
    return idx;
}

}  // namespace oracle
