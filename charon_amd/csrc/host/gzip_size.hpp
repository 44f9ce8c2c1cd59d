// gzip_size.hpp -- the SIZE of gzip::compress(sequence) without producing the bytes.
//
// get_compression_ratio (src/utils.cpp:114-124) needs only the length of the gzip member that gzip-hpp @7546b35 produces with
// deflateInit2(Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) and one deflate(Z_FINISH).  With a four-letter
// alphabet zlib spends its time walking saturated hash chains (one dependent load per candidate).  This restates zlib's level-6
// algorithm (deflate_slow + longest_match + the block-type decision of _tr_flush_block; zlib 1.2.11 is the version in the image,
// the algorithm has been stable across 1.2.x / 1.3.x) on per-hash-class occurrence ARRAYS instead of linked chains, so the
// candidate loop streams through memory, and it counts bits instead of writing them.  Every decision zlib takes is taken
// identically (same candidates in the same order, same chain and lazy-match limits, same Huffman code lengths), so the size
// is bit-exact; tests/test_cli_cpu.py and tools/gzip_size_check.cpp compare it with the linked zlib on millions of inputs, and
// the front end re-checks a few known strings at start-up and falls back to zlib itself if they ever disagree.
//
// Fast path (inputs made of A, C, G, T, N only -- what sequence_to_string produces): longest_match is a pure function of the
// candidate set E = the first `chain_length` same-hash positions inside the window, in recency order: it returns the most recent
// candidate of maximal match length, or the first one that reaches nice_match.  Candidates that share at least six bytes with the
// current string are exactly the earlier occurrences of its 6-mer, so they are enumerated through a 6-mer occurrence chain (about
// two per search on DNA instead of 128 same-trigram candidates), membership in E being decided from the position's rank inside
// its hash class; if there is none, the answer is the most recent occurrence of the 5-, 4- or 3-mer inside E.  Same result, a
// fraction of the work.
//
// Scope: A/C/G/T/N inputs of any length up to 1 GiB (zlib's window slide -- fill_window / slide_hash -- is emulated: the window
// base advances by w_size whenever strstart passes w_size + MAX_DIST, positions at or below the base are NIL); other 7-bit inputs
// up to GzipSizer::MAX_BYTES through the generic candidate loop.  size() returns 0 for anything else and the caller uses zlib.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace gzsize {

class GzipSizer {
public:
    static const size_t MAX_BYTES = 60000;          // generic (non-DNA) inputs: the whole input must fit zlib's 64 KiB window without a slide
    static const size_t MAX_DNA_BYTES = 1u << 30;   // A/C/G/T/N inputs: the window slide is emulated

    // total bytes of the gzip member (10-byte header + deflate stream + 8-byte trailer); 0 if the input is out of scope
    uint32_t size(const uint8_t *data, size_t n) {
        if (n == 0 || n > MAX_DNA_BYTES) return 0;
        buf_.assign(n + PAD, 0);
        std::memcpy(buf_.data(), data, n);
        return size_padded(buf_.data(), n);
    }
    // same, for a caller whose buffer already has PAD zero bytes behind the n data bytes (no copy)
    static const size_t PAD = 258 + 16;
    uint32_t size_padded(const uint8_t *data, size_t n) {
        if (n == 0 || n > MAX_DNA_BYTES) return 0;
        in_ = data;
        if (!prepare_dna(n)) {
            if (n > MAX_BYTES) return 0;  // the generic path does not emulate the window slide
            for (size_t i = 0; i < n; ++i)
                if (data[i] & 0x80) return 0;
            prepare(n);
        }
        run(n);
        return (uint32_t)(18 + ((total_bits_ + 7) >> 3));
    }

    // The same size from the symbol tallies of a ONE-block deflate_slow run that something else did (the GPU: k_gzip_tally,
    // include/charon_hip.h chn_batch.gzip_tallies): lfreq[286] literal/length code frequencies WITHOUT the end-of-block symbol's
    // count being trusted (it is set here), dfreq[30] distance code frequencies, n input bytes.
    uint32_t size_from_tallies(const uint16_t *lfreq, const uint16_t *dfreq, size_t n) {
        ensure_tables();
        total_bits_ = 0;
        init_block();
        for (int i = 0; i < L_CODES; ++i) lfreq_[i] = lfreq[i];
        lfreq_[END_BLOCK] = 1;
        for (int i = 0; i < D_CODES; ++i) dfreq_[i] = dfreq[i];
        flush_block(n, true, true);
        return (uint32_t)(18 + ((total_bits_ + 7) >> 3));
    }

private:
    enum {
        MIN_MATCH = 3, MAX_MATCH = 258, W_SIZE = 32768, MIN_LOOKAHEAD = MAX_MATCH + MIN_MATCH + 1, MAX_DIST = W_SIZE - MIN_LOOKAHEAD,
        TOO_FAR = 4096, GOOD_MATCH = 8, MAX_LAZY = 16, NICE_MATCH = 128, MAX_CHAIN = 128,  // configuration_table[6], deflate_slow
        LIT_BUFSIZE = 1 << (8 + 6),  // memLevel 8
        L_CODES = 286, D_CODES = 30, BL_CODES = 19, LITERALS = 256, END_BLOCK = 256, MAX_BITS = 15, MAX_BL_BITS = 7,
        HEAP_SIZE = 2 * L_CODES + 1, REP_3_6 = 16, REPZ_3_10 = 17, REPZ_11_138 = 18, HASH_SIZE = 32768
    };

    // ---- input + occurrence arrays -----------------------------------------------------------------------------------
    std::vector<uint8_t> buf_;        // input + zero padding (zlib zeroes WIN_INIT bytes behind the data, so compares past the end see 0)
    std::vector<uint32_t> occ_;       // positions that enter the dictionary, grouped by hash class, ascending inside a class
    std::vector<uint32_t> rank_;      // rank_[p] = index of p in occ_
    std::vector<uint32_t> cls_lo_;    // first index of p's class in occ_ (per position, to find the class start)
    std::vector<uint32_t> count_;     // per hash value (HASH_SIZE), kept all-zero between calls
    std::vector<uint16_t> used_;

    static uint32_t hash3(const uint8_t *p) {  // UPDATE_HASH x3 with hash_shift 5, hash_bits 15: exactly the last three bytes
        return (((uint32_t)p[0] << 10) ^ ((uint32_t)p[1] << 5) ^ (uint32_t)p[2]) & (HASH_SIZE - 1);
    }

    const uint8_t *in_ = nullptr;     // the (zero-padded) input of the current call
    void prepare(size_t n) {
        if (count_.empty()) count_.assign(HASH_SIZE, 0);
        const size_t m = n >= MIN_MATCH ? n - MIN_MATCH + 1 : 0;  // positions 0 .. n-3 are inserted (INSERT_STRING needs 3 bytes)
        occ_.resize(m);
        rank_.resize(m);
        cls_lo_.resize(m);
        used_.clear();
        std::vector<uint16_t> &h = hash_;
        h.resize(m);
        for (size_t p = 0; p < m; ++p) {
            const uint32_t v = hash3(in_ + p);
            h[p] = (uint16_t)v;
            if (count_[v]++ == 0) used_.push_back((uint16_t)v);
        }
        uint32_t run = 0;
        for (uint16_t v : used_) { const uint32_t c = count_[v]; count_[v] = run; run += c; }  // count_ becomes the class cursor
        start_.resize(used_.size());
        for (size_t i = 0; i < used_.size(); ++i) start_[i] = count_[used_[i]];
        for (size_t p = 0; p < m; ++p) {
            const uint32_t v = h[p], r = count_[v]++;
            occ_[r] = (uint32_t)p;
            rank_[p] = r;
        }
        // class start per position: walk the classes again (count_ now holds the class END)
        for (size_t i = 0; i < used_.size(); ++i) {
            const uint32_t lo = start_[i], hi = count_[used_[i]];
            for (uint32_t r = lo; r < hi; ++r) cls_lo_[occ_[r]] = lo;
            count_[used_[i]] = 0;  // leave the table clean for the next call
        }
    }
    std::vector<uint16_t> hash_;
    std::vector<uint32_t> start_;
    // fast path: rank of each position inside its hash class and the previous occurrence (+1, 0 = none) of the 3/4/5/6-mer starting
    // there.  The tables are epoch-tagged (high 16 bits) so nothing has to be cleared between reads; positions fit 16 bits.
    bool dna_ = false;
    std::vector<uint32_t> rank16_, prev3_, prev4_, prev5_, prev6_;
    std::vector<uint64_t> cnt_, last3_, last4_, last5_, last6_;
    uint64_t epoch_ = 0;
    int8_t dna_code_[256];
    bool dna_tab_ = false;
    bool prepare_dna(size_t n) {
        if (!dna_tab_) {
            std::memset(dna_code_, -1, sizeof dna_code_);
            dna_code_[(int)'A'] = 0; dna_code_[(int)'C'] = 1; dna_code_[(int)'G'] = 2; dna_code_[(int)'T'] = 3; dna_code_[(int)'N'] = 4;
            // base-5 k-mer keys keep the tables small (125 / 625 / 3 125 / 15 625 entries): they stay cache-resident for short reads.
            // The 15-bit zlib hash separates all 125 DNA trigrams (checked here), so a hash class IS a trigram class.
            bool distinct = true;
            {
                std::vector<uint8_t> seen(HASH_SIZE, 0);
                const char *al = "ACGTN";
                for (int a = 0; a < 5; ++a) for (int b = 0; b < 5; ++b) for (int c = 0; c < 5; ++c) {
                    const uint8_t t[3] = {(uint8_t)al[a], (uint8_t)al[b], (uint8_t)al[c]};
                    uint8_t &sv = seen[hash3(t)];
                    if (sv) distinct = false;
                    sv = 1;
                }
            }
            dna_ok_ = distinct;
            cnt_.assign(125, 0); last3_.assign(125, 0); last4_.assign(625, 0); last5_.assign(3125, 0); last6_.assign(15625, 0);
            dna_tab_ = true;
        }
        dna_ = false;
        if (!dna_ok_) return false;
        ++epoch_;  // tables are tagged with the call number (high 32 bits): nothing is cleared between reads
        const uint64_t tag = epoch_ << 32;
        if (rank16_.size() < n + 8) { rank16_.resize(n + 8); prev3_.resize(n + 8); prev4_.resize(n + 8); prev5_.resize(n + 8); prev6_.resize(n + 8); }
        const uint8_t *d = in_;
        uint32_t k3 = 0, k4 = 0, k5 = 0, k6 = 0;  // base-5 keys of the 3/4/5/6-mers ENDING at the current byte
        uint32_t hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // the last codes, by position & 7 (zero before the start)
        for (size_t e = 0; e < n; ++e) {
            const int ci = dna_code_[d[e]];
            if (ci < 0) return false;
            const uint32_t c = (uint32_t)ci;
            // the digit that leaves a window of length L is the code at e - L
            k3 = (k3 - hist[(e + 5) & 7] * 25u) * 5u + c;
            k4 = (k4 - hist[(e + 4) & 7] * 125u) * 5u + c;
            k5 = (k5 - hist[(e + 3) & 7] * 625u) * 5u + c;
            k6 = (k6 - hist[(e + 2) & 7] * 3125u) * 5u + c;
            hist[e & 7] = c;
            if (e >= 2) {
                const size_t p = e - 2;
                uint64_t &ct = cnt_[k3];
                if ((ct >> 32) != epoch_) ct = tag;
                rank16_[p] = (uint32_t)ct;
                ++ct;
                uint64_t &l3 = last3_[k3];
                prev3_[p] = (l3 >> 32) == epoch_ ? (uint32_t)l3 : 0u;
                l3 = tag | (uint64_t)(p + 1);
                if (e >= 3) {
                    uint64_t &l4 = last4_[k4];
                    prev4_[p - 1] = (l4 >> 32) == epoch_ ? (uint32_t)l4 : 0u;
                    l4 = tag | (uint64_t)p;
                }
                if (e >= 4) {
                    uint64_t &l5 = last5_[k5];
                    prev5_[p - 2] = (l5 >> 32) == epoch_ ? (uint32_t)l5 : 0u;
                    l5 = tag | (uint64_t)(p - 1);
                }
                if (e >= 5) {
                    uint64_t &l6 = last6_[k6];
                    prev6_[p - 3] = (l6 >> 32) == epoch_ ? (uint32_t)l6 : 0u;
                    l6 = tag | (uint64_t)(p - 2);
                }
            }
        }
        // k-mers that do not fit the input have no occurrence chain
        for (size_t p = n >= 2 ? n - 2 : 0; p < n; ++p) prev3_[p] = 0;
        for (size_t p = n >= 3 ? n - 3 : 0; p < n; ++p) prev4_[p] = 0;
        for (size_t p = n >= 4 ? n - 4 : 0; p < n; ++p) prev5_[p] = 0;
        for (size_t p = n >= 5 ? n - 5 : 0; p < n; ++p) prev6_[p] = 0;
        dna_ = true;
        return true;
    }
    bool dna_ok_ = false;

    static inline uint16_t ld16(const uint8_t *p) { uint16_t v; std::memcpy(&v, p, 2); return v; }
    static inline uint64_t ld64(const uint8_t *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

    // ---- longest_match (deflate.c), same candidates in the same order ------------------------------------------------
    uint32_t match_start_ = 0;
    uint32_t base_ = 0;  // absolute position of the window start (multiples of w_size; 0 until the first slide)
    uint32_t longest_match(uint32_t strstart, uint32_t lookahead, uint32_t prev_length) {
        uint32_t chain_length = MAX_CHAIN;
        uint32_t best_len = prev_length;
        uint32_t nice_match = NICE_MATCH;
        const uint32_t limit = strstart > (uint32_t)MAX_DIST ? strstart - (uint32_t)MAX_DIST : 0;
        const uint8_t *scan = in_ + strstart;
        if (prev_length >= GOOD_MATCH) chain_length >>= 2;
        if (nice_match > lookahead) nice_match = lookahead;
        uint16_t scan_end = ld16(scan + best_len - 1);
        const uint16_t scan_start = ld16(scan);
        const uint32_t lo = cls_lo_[strstart];
        uint32_t r = rank_[strstart];  // candidates are occ_[r-1], occ_[r-2], ... (the caller checked that the first one is valid)
        bool first = true;
        while (r > lo) {
            const uint32_t cur = occ_[--r];
            if (!first && cur <= limit) break;  // `(cur_match = prev[...]) > limit`; position 0 is NIL and never above a limit
            first = false;
            const uint8_t *match = in_ + cur;
            if (ld16(match + best_len - 1) == scan_end && ld16(match) == scan_start) {
                // bytes 0,1 equal and equal hash => byte 2 equal (7-bit input); compare on from byte 3, at most up to MAX_MATCH
                uint32_t len = 3;
                while (len < (uint32_t)MAX_MATCH) {
                    const uint64_t x = ld64(scan + len) ^ ld64(match + len);
                    if (x) { len += (uint32_t)(__builtin_ctzll(x) >> 3); break; }
                    len += 8;
                }
                if (len > (uint32_t)MAX_MATCH) len = MAX_MATCH;
                if (len > best_len) {
                    match_start_ = cur;
                    best_len = len;
                    if (len >= nice_match) break;
                    scan_end = ld16(scan + best_len - 1);
                }
            }
            if (--chain_length == 0) break;
        }
        return best_len <= lookahead ? best_len : lookahead;
    }

    // the same function of the candidate set, evaluated through the k-mer occurrence chains (see the header comment)
    uint32_t longest_match_dna(uint32_t strstart, uint32_t lookahead, uint32_t prev_length) {
        const uint32_t K = prev_length >= (uint32_t)GOOD_MATCH ? (uint32_t)MAX_CHAIN >> 2 : (uint32_t)MAX_CHAIN;
        const uint32_t nice_match = (uint32_t)NICE_MATCH > lookahead ? lookahead : (uint32_t)NICE_MATCH;
        // limit = strstart > MAX_DIST ? strstart - MAX_DIST : NIL, in window-relative positions; NIL is the window base
        const uint32_t limit = strstart - base_ > (uint32_t)MAX_DIST ? strstart - (uint32_t)MAX_DIST : base_;
        const uint32_t rs = rank16_[strstart];
        uint32_t best_len = prev_length;
        const uint8_t *scan = in_ + strstart;
        auto in_e = [&](uint32_t c) {
            const uint32_t d = rs - rank16_[c];  // 1 = the head of the chain
            if (d > K) return false;
            // the head was vetted by the caller (not NIL, within MAX_DIST); the others must lie above the limit
            return d == 1 ? (c > base_ && strstart - c <= (uint32_t)MAX_DIST) : c > limit;
        };
        bool any6 = false;
        for (uint32_t c1 = prev6_[strstart]; c1 != 0; c1 = prev6_[c1 - 1]) {
            const uint32_t cur = c1 - 1;
            if (!in_e(cur)) break;
            any6 = true;
            const uint8_t *match = in_ + cur;
            uint32_t len = 6;
            while (len < (uint32_t)MAX_MATCH) {
                const uint64_t x = ld64(scan + len) ^ ld64(match + len);
                if (x) { len += (uint32_t)(__builtin_ctzll(x) >> 3); break; }
                len += 8;
            }
            if (len > (uint32_t)MAX_MATCH) len = MAX_MATCH;
            if (len > best_len) {
                match_start_ = cur;
                best_len = len;
                if (len >= nice_match) break;
            }
        }
        if (!any6) {
            uint32_t c1;
            if ((c1 = prev5_[strstart]) != 0 && in_e(c1 - 1)) { if (5 > best_len) { match_start_ = c1 - 1; best_len = 5; } }
            else if ((c1 = prev4_[strstart]) != 0 && in_e(c1 - 1)) { if (4 > best_len) { match_start_ = c1 - 1; best_len = 4; } }
            else if ((c1 = prev3_[strstart]) != 0 && in_e(c1 - 1)) { if (3 > best_len) { match_start_ = c1 - 1; best_len = 3; } }
        }
        return best_len <= lookahead ? best_len : lookahead;
    }

    // ---- trees.c: tallies and block sizes --------------------------------------------------------------------------------
    uint32_t lfreq_[HEAP_SIZE], dfreq_[2 * D_CODES + 1], blfreq_[2 * BL_CODES + 1];
    uint32_t last_lit_ = 0;
    uint64_t total_bits_ = 0, opt_len_ = 0, static_len_ = 0;
    uint8_t length_code_[256], dist_code_[512];
    bool tables_ = false;

    static const int *extra_lbits() {
        static const int t[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        return t;
    }
    static const int *extra_dbits() {
        static const int t[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        return t;
    }
    static const int *extra_blbits() {
        static const int t[19] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 7};
        return t;
    }
    void init_tables() {  // tr_static_init
        int length = 0, code;
        for (code = 0; code < 28; ++code)
            for (int n = 0; n < (1 << extra_lbits()[code]); ++n) length_code_[length++] = (uint8_t)code;
        length_code_[length - 1] = (uint8_t)code;
        int dist = 0;
        for (code = 0; code < 16; ++code)
            for (int n = 0; n < (1 << extra_dbits()[code]); ++n) dist_code_[dist++] = (uint8_t)code;
        dist >>= 7;
        for (; code < D_CODES; ++code)
            for (int n = 0; n < (1 << (extra_dbits()[code] - 7)); ++n) dist_code_[256 + dist++] = (uint8_t)code;
        tables_ = true;
    }
    void init_block() {
        std::memset(lfreq_, 0, sizeof lfreq_);
        std::memset(dfreq_, 0, sizeof dfreq_);
        std::memset(blfreq_, 0, sizeof blfreq_);
        lfreq_[END_BLOCK] = 1;
        opt_len_ = static_len_ = 0;
        last_lit_ = 0;
    }
    bool tally_lit(uint8_t c) {
        lfreq_[c]++;
        return ++last_lit_ == LIT_BUFSIZE - 1;
    }
    bool tally_dist(uint32_t dist, uint32_t lc) {  // dist = match distance, lc = match length - MIN_MATCH
        lfreq_[length_code_[lc] + LITERALS + 1]++;
        --dist;
        dfreq_[dist < 256 ? dist_code_[dist] : dist_code_[256 + (dist >> 7)]]++;
        return ++last_lit_ == LIT_BUFSIZE - 1;
    }

    struct TreeDesc {
        uint32_t *freq;       // [2*elems+1]
        const uint8_t *slen;  // static code lengths or null
        const int *extra;
        int extra_base, elems, max_length, max_code;
    };
    uint16_t len_[HEAP_SIZE], dad_[HEAP_SIZE];
    uint16_t llen_[HEAP_SIZE], dlen_[2 * D_CODES + 1], bllen_[2 * BL_CODES + 1];
    int heap_[HEAP_SIZE], heap_len_ = 0, heap_max_ = 0;
    uint8_t depth_[HEAP_SIZE];
    uint16_t bl_count_[MAX_BITS + 1];

    bool smaller(const uint32_t *f, int n, int m) const { return f[n] < f[m] || (f[n] == f[m] && depth_[n] <= depth_[m]); }
    void pqdownheap(const uint32_t *f, int k) {
        const int v = heap_[k];
        int j = k << 1;
        while (j <= heap_len_) {
            if (j < heap_len_ && smaller(f, heap_[j + 1], heap_[j])) j++;
            if (smaller(f, v, heap_[j])) break;
            heap_[k] = heap_[j];
            k = j;
            j <<= 1;
        }
        heap_[k] = v;
    }
    void build_tree(TreeDesc &d, uint16_t *outlen) {
        uint32_t *f = d.freq;
        const int elems = d.elems;
        int max_code = -1, node;
        heap_len_ = 0;
        heap_max_ = HEAP_SIZE;
        for (int n = 0; n < elems; ++n) {
            if (f[n] != 0) { heap_[++heap_len_] = max_code = n; depth_[n] = 0; }
            else len_[n] = 0;
        }
        while (heap_len_ < 2) {
            node = heap_[++heap_len_] = (max_code < 2 ? ++max_code : 0);
            f[node] = 1;
            depth_[node] = 0;
            opt_len_--;
            if (d.slen) static_len_ -= d.slen[node];
        }
        d.max_code = max_code;
        for (int n = heap_len_ / 2; n >= 1; --n) pqdownheap(f, n);
        node = elems;
        do {
            const int n = heap_[1];
            heap_[1] = heap_[heap_len_--];
            pqdownheap(f, 1);
            const int m = heap_[1];
            heap_[--heap_max_] = n;
            heap_[--heap_max_] = m;
            f[node] = f[n] + f[m];
            depth_[node] = (uint8_t)((depth_[n] >= depth_[m] ? depth_[n] : depth_[m]) + 1);
            dad_[n] = dad_[m] = (uint16_t)node;
            heap_[1] = node++;
            pqdownheap(f, 1);
        } while (heap_len_ >= 2);
        heap_[--heap_max_] = heap_[1];
        // gen_bitlen
        int h, overflow = 0;
        for (int bits = 0; bits <= MAX_BITS; ++bits) bl_count_[bits] = 0;
        len_[heap_[heap_max_]] = 0;
        for (h = heap_max_ + 1; h < HEAP_SIZE; ++h) {
            const int n = heap_[h];
            int bits = len_[dad_[n]] + 1;
            if (bits > d.max_length) { bits = d.max_length; overflow++; }
            len_[n] = (uint16_t)bits;
            if (n > max_code) continue;
            bl_count_[bits]++;
            int xbits = 0;
            if (n >= d.extra_base) xbits = d.extra[n - d.extra_base];
            opt_len_ += (uint64_t)f[n] * (unsigned)(bits + xbits);
            if (d.slen) static_len_ += (uint64_t)f[n] * (unsigned)(d.slen[n] + xbits);
        }
        if (overflow != 0) {
            do {
                int bits = d.max_length - 1;
                while (bl_count_[bits] == 0) bits--;
                bl_count_[bits]--;
                bl_count_[bits + 1] += 2;
                bl_count_[d.max_length]--;
                overflow -= 2;
            } while (overflow > 0);
            for (int bits = d.max_length; bits != 0; --bits) {
                int n = bl_count_[bits];
                while (n != 0) {
                    const int m = heap_[--h];
                    if (m > max_code) continue;
                    if ((unsigned)len_[m] != (unsigned)bits) {
                        opt_len_ += ((uint64_t)bits - len_[m]) * f[m];
                        len_[m] = (uint16_t)bits;
                    }
                    n--;
                }
            }
        }
        for (int n = 0; n <= max_code; ++n) outlen[n] = len_[n];
        for (int n = max_code + 1; n < elems; ++n) outlen[n] = 0;
    }
    void scan_tree(uint16_t *tlen, int max_code) {
        int prevlen = -1, curlen, nextlen = tlen[0], count = 0, max_count = 7, min_count = 4;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        tlen[max_code + 1] = (uint16_t)0xffff;  // guard
        for (int n = 0; n <= max_code; ++n) {
            curlen = nextlen;
            nextlen = tlen[n + 1];
            if (++count < max_count && curlen == nextlen) continue;
            else if (count < min_count) blfreq_[curlen] += (uint32_t)count;
            else if (curlen != 0) {
                if (curlen != prevlen) blfreq_[curlen]++;
                blfreq_[REP_3_6]++;
            } else if (count <= 10) blfreq_[REPZ_3_10]++;
            else blfreq_[REPZ_11_138]++;
            count = 0;
            prevlen = curlen;
            if (nextlen == 0) { max_count = 138; min_count = 3; }
            else if (curlen == nextlen) { max_count = 6; min_count = 3; }
            else { max_count = 7; min_count = 4; }
        }
    }
    uint8_t static_llen_[L_CODES + 2], static_dlen_[D_CODES];

    // _tr_flush_block: adds the bits of this block to total_bits_
    void flush_block(uint64_t stored_len, bool last, bool buf_available = true) {
        TreeDesc ld = {lfreq_, static_llen_, extra_lbits(), LITERALS + 1, L_CODES, MAX_BITS, -1};
        TreeDesc dd = {dfreq_, static_dlen_, extra_dbits(), 0, D_CODES, MAX_BITS, -1};
        TreeDesc bd = {blfreq_, nullptr, extra_blbits(), 0, BL_CODES, MAX_BL_BITS, -1};
        build_tree(ld, llen_);
        build_tree(dd, dlen_);
        scan_tree(llen_, ld.max_code);
        scan_tree(dlen_, dd.max_code);
        build_tree(bd, bllen_);
        static const uint8_t bl_order[BL_CODES] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int max_blindex;
        for (max_blindex = BL_CODES - 1; max_blindex >= 3; --max_blindex)
            if (bllen_[bl_order[max_blindex]] != 0) break;
        opt_len_ += 3 * ((uint64_t)max_blindex + 1) + 5 + 5 + 4;
        uint64_t opt_lenb = (opt_len_ + 3 + 7) >> 3;
        const uint64_t static_lenb = (static_len_ + 3 + 7) >> 3;
        if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
        if (stored_len + 4 <= opt_lenb && buf_available) {
            // _tr_stored_block: 3 header bits, pad to a byte, LEN + NLEN, the bytes
            total_bits_ += 3;
            total_bits_ = (total_bits_ + 7) & ~7ULL;
            total_bits_ += 32 + 8 * stored_len;
        } else if (static_lenb == opt_lenb) {
            total_bits_ += 3 + static_len_;
        } else {
            total_bits_ += 3 + opt_len_;
        }
        init_block();
        if (last) total_bits_ = (total_bits_ + 7) & ~7ULL;  // bi_windup
    }

    // ---- deflate_slow ----------------------------------------------------------------------------------------------------
    void ensure_tables() {
        if (!tables_) {
            init_tables();
            for (int n = 0; n <= 143; ++n) static_llen_[n] = 8;
            for (int n = 144; n <= 255; ++n) static_llen_[n] = 9;
            for (int n = 256; n <= 279; ++n) static_llen_[n] = 7;
            for (int n = 280; n < L_CODES + 2; ++n) static_llen_[n] = 8;
            for (int n = 0; n < D_CODES; ++n) static_dlen_[n] = 5;
        }
    }
    void run(size_t n_in) {
        ensure_tables();
        const uint32_t n = (uint32_t)n_in;
        total_bits_ = 0;
        init_block();
        uint32_t strstart = 0, block_start = 0, match_length = MIN_MATCH - 1, prev_length, prev_match;
        bool match_available = false;
        match_start_ = 0;
        base_ = 0;
        uint32_t loaded_end = n < 2u * W_SIZE ? n : 2u * W_SIZE;  // the first fill_window reads as much as the window holds
        for (;;) {
            if (loaded_end - strstart < (uint32_t)MIN_LOOKAHEAD) {
                // fill_window: slide the window by w_size once strstart has passed w_size + MAX_DIST (even when no input is left),
                // then read as much as fits
                if (strstart - base_ >= (uint32_t)(W_SIZE + MAX_DIST)) base_ += W_SIZE;
                const uint64_t cap = (uint64_t)base_ + 2u * W_SIZE;
                loaded_end = n < cap ? n : (uint32_t)cap;
            }
            const uint32_t lookahead = loaded_end - strstart;
            if (lookahead == 0) break;
            uint32_t hash_head = 0;  // NIL
            if (lookahead >= (uint32_t)MIN_MATCH) {
                if (dna_) {
                    // zlib only asks whether the class has an earlier member and how far the most recent one is; for DNA letters the
                    // 15-bit hash separates all 125 trigrams, so that member is the previous occurrence of the trigram
                    hash_head = prev3_[strstart] ? prev3_[strstart] - 1 : 0;
                    if (hash_head <= base_) hash_head = 0;  // slide_hash turned everything at or below the window base into NIL
                } else {
                    const uint32_t r = rank_[strstart];
                    if (r > cls_lo_[strstart]) hash_head = occ_[r - 1];  // the previous position of the same hash class (0 == NIL, as in zlib)
                }
            }
            prev_length = match_length;
            prev_match = match_start_;
            match_length = MIN_MATCH - 1;
            if (hash_head != 0 && prev_length < (uint32_t)MAX_LAZY && strstart - hash_head <= (uint32_t)MAX_DIST) {  // hash_head != NIL
                match_length = dna_ ? longest_match_dna(strstart, lookahead, prev_length) : longest_match(strstart, lookahead, prev_length);
                if (match_length <= 5 && (match_length == (uint32_t)MIN_MATCH && strstart - match_start_ > (uint32_t)TOO_FAR)) match_length = MIN_MATCH - 1;
            }
            if (prev_length >= (uint32_t)MIN_MATCH && match_length <= prev_length) {
                const bool bflush = tally_dist(strstart - 1 - prev_match, prev_length - MIN_MATCH);
                strstart += prev_length - 1;  // the strings up to the end of the match enter the dictionary (implicit here)
                match_available = false;
                match_length = MIN_MATCH - 1;
                if (bflush) { flush_block(strstart - block_start, false, block_start >= base_); block_start = strstart; }
            } else if (match_available) {
                const bool bflush = tally_lit(in_[strstart - 1]);
                if (bflush) { flush_block(strstart - block_start, false, block_start >= base_); block_start = strstart; }
                strstart++;
            } else {
                match_available = true;
                strstart++;
            }
        }
        if (match_available) tally_lit(in_[strstart - 1]);
        flush_block(strstart - block_start, true, block_start >= base_);
    }
};

}  // namespace gzsize
