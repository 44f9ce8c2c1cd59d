// libcharon_hip.so -- MI355X (gfx950 / CDNA4) implementation of the per-read classification path of
// `charon dehost` behind the C ABI of include/charon_hip.h.
//
// Kernel chain per batch (all on one HIP stream, no host round trip):
//   k_len_bucket_*      order reads by length class so that a wavefront holds reads of similar length
//   k_minimise_probe    ONE LANE PER READ: rolls the canonical base-5 k-mer hash (seqan3 minimiser_hash
//                       semantics, src/dehost_main.cpp:317-318,367), runs the exact sequential window-minimum
//                       emission rule (so ties/homopolymers are exact by construction), compacts emitted
//                       minimisers of the 64 reads of the wavefront into an LDS queue, and every 64 queued
//                       minimisers does one full-width probe round: 64 lanes x h gathers of W words from the
//                       HBM-resident interleaved Bloom filter (bulk_contains, src/dehost_main.cpp:368), AND,
//                       then either accumulates per-category hit / unique-hit counters in LDS (FUSED: every
//                       category owns one bin, C <= 8) or appends the bit-row to the read's row list in HBM.
//   k_count_rows        (general layouts) one wavefront per read: per-bin totals -> first max bin per
//                       category -> unique hits  (ReadEntry::get_counts, include/read_entry.hpp:92-138)
//   k_model_call        one lane per read: KDE/dexp probability + call_host / call_category
//                       (include/classify_stats.hpp:242-252,370-389; include/read_entry.hpp:157-279)
// No MFMA: the path is integer/bit work bound by random 8-16 byte gathers from HBM.
//
// This file is written for gfx950 only.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "charon_hip.h"
#include "default_kde.inc"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            return fail(_e == hipErrorOutOfMemory ? CHN_E_NOMEM : CHN_E_HIP,                                 \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                                  \
    } while (0)

extern "C" const char *chn_last_error(void) { return g_err.c_str(); }
extern "C" const char *chn_version(void) { return "charon_hip 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#define WAVE 64
#define QCAP 128  // LDS minimiser queue entries per wavefront (power of two, >= 2*WAVE)

__device__ __constant__ uint64_t c_ibf_seeds[5] = {13572355802537770549ULL, 13043817825332782213ULL,
                                                   10650232656628343401ULL, 16499269484942379435ULL,
                                                   4893150838803335377ULL};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t t = (uint64_t)__shfl_xor((long long)v, o);
        v = t > v ? t : v;
    }
    return v;
}

// Loads whose completion the KERNEL tracks instead of the compiler (cdna_hip_programming.md 5.7): hipcc does not see a
// load inside an asm statement, so it inserts no s_waitcnt for it -- the caller must wait before touching the result.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4_t asm_load_b128(const void *p) {
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ u32x2_t asm_load_b64(const void *p) {
    u32x2_t v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// seqan3 interleaved_bloom_filter::hash_and_fit (fastrange), SURVEY App. A.3
__device__ __forceinline__ uint64_t hash_and_fit_row(uint64_t x, uint64_t seed, uint64_t S, uint32_t shift) {
    x *= seed;
    x ^= x >> shift;
    x *= 11400714819323198485ULL;
    return __umul64hi(x, S);
}

// counter-based PRNG (splitmix64 finaliser) shared by the synthetic-workload kernels
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// ------------------------------------------------------------------------------------------------
// k_minimise_probe
// ------------------------------------------------------------------------------------------------
enum { MODE_FUSED = 0, MODE_ROWS = 1, MODE_EMPLACE = 2, MODE_LIST = 3 };  // LIST: log the minimiser values themselves (row-sharded mode)

struct K1Args {
    const uint64_t *words;   // IBF shard, word (row - row_begin) * W + b
    uint64_t *words_rw;      // MODE_EMPLACE
    uint64_t S, row_begin, row_end, seed, powk1;
    uint32_t shift, h, k, wn, n_reads, nseg, B, C;
    uint64_t b2c_packed;     // MODE_FUSED: category of bin b in byte b
    const uint32_t *bases, *nmask;
    const uint64_t *off1, *off2;
    const uint32_t *len1, *len2;
    const uint32_t *order;
    uint32_t *num_hashes, *counts, *unique;
    uint64_t *rows;          // MODE_ROWS: per-wavefront row log, entry e of wavefront g at rows[(wave_base[g] + e) * W]
    uint32_t *rowlog;        // MODE_ROWS: compact entry (see encode_row) of every log entry; MODE_LIST: owner lane
    const uint64_t *wave_base;
    uint32_t *wave_count;    // MODE_ROWS: entries written by wavefront g
    uint32_t nt_probes;      // non-temporal row gathers (index much larger than the Infinity Cache)
    const uint8_t *read_bin; // MODE_EMPLACE: target bin of "read" (genome chunk) r
    uint32_t ablate;         // diagnostics only (CHN_ABLATE env): 1 = skip the gathers, 2 = skip hash+gathers
};

// Compact row-log entry: bits 0-1 number of set bins (0..3), bits 2-9 / 10-17 / 18-25 their indices, bits 26-31 the owner
// lane.  A row with more than three set bins is rare (true bins + ~1 % false positives per bin): it is flagged by
// count = 0 with a non-zero first index field and its W full words are stored at the same entry of the full-row buffer.
#define ROWLOG_ESCAPE 4u  // count 0, idx0 field = 1
template <int W>
__device__ __forceinline__ uint32_t encode_row(const uint64_t *acc, uint32_t B, uint32_t owner, bool &escaped) {
    uint64_t m[W];
    uint32_t pc = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        m[w] = acc[w];
        if (w == W - 1 && (B & 63u)) m[w] &= (1ULL << (B & 63u)) - 1;  // technical bins >= B never count
        pc += (uint32_t)__popcll(m[w]);
    }
    escaped = pc > 3;
    if (escaped) return ROWLOG_ESCAPE | (owner << 26);
    uint32_t e = pc | (owner << 26), slot = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        uint64_t x = m[w];
        while (x) {
            const uint32_t b = (uint32_t)w * 64 + (uint32_t)__ffsll((long long)x) - 1;
            x &= x - 1;
            e |= b << (2 + 8 * slot);
            ++slot;
        }
    }
    return e;
}

template <int W>
__device__ __forceinline__ void load_row_and(const uint64_t *p, uint64_t *acc) {
    if (W == 1) {
        acc[0] &= p[0];
    } else if (W == 2) {
        ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p);
        acc[0] &= v.x; acc[1] &= v.y;
    } else if (W == 3) {
        acc[0] &= p[0]; acc[1] &= p[1]; acc[2] &= p[2];
    } else {
        ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(p);
        ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(p + 2);
        acc[0] &= a.x; acc[1] &= a.y; acc[2] &= b.x; acc[3] &= b.y;
    }
}

// Sliding-window minimum without rescans.  Value indices are cut into blocks of wn; slot u of the per-lane LDS
// ring holds, for u <= t (t = offset of the newest value in its block), the raw values of the current block and,
// for u > t, the rightmost suffix minimum S[u] of the PREVIOUS block (value in `ring`, offset in `spos`).  The
// rightmost minimum of the window ending at offset t is then combine(S[t+1], running prefix minimum), and the
// suffix minima are produced in place by one backward pass at every block end (all lanes are at the same t).
//
// Probe rounds are software-pipelined: a round's h*W-word gathers are issued into registers and only consumed
// (AND + accumulate / row store) when the next round is due ~11 bases later, so the HBM latency of the random
// gathers overlaps the hash rolling of the same wavefront.  Bases are fetched 64 at a time (one dwordx4 per lane)
// one chunk ahead, so the only vector-memory wait inside the base loop sits at a chunk boundary.
template <int W, int MODE, int WN_T>
__global__ __launch_bounds__(WAVE) void k_minimise_probe(const K1Args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t wn = WN_T ? (uint32_t)WN_T : a.wn;
    // LDS carve-up (one wavefront per workgroup)
    uint64_t *ring = reinterpret_cast<uint64_t *>(smem);                      // [wn][64] per-lane window store
    uint64_t *qv = ring + (size_t)wn * WAVE;                                  // [QCAP] queued minimiser values
    uint32_t *qm = reinterpret_cast<uint32_t *>(qv + QCAP);                   // [QCAP] owner lane | idx << 6
    uint64_t *rbase = reinterpret_cast<uint64_t *>(qm + QCAP);                // [64] MODE_ROWS row base / EMPLACE bin
    uint32_t *cnt = reinterpret_cast<uint32_t *>(rbase + WAVE);               // MODE_FUSED [C][64]
    uint32_t *unq = cnt + (MODE == MODE_FUSED ? a.C * WAVE : 0);              // MODE_FUSED [C][64]
    uint8_t *spos = reinterpret_cast<uint8_t *>(unq + (MODE == MODE_FUSED ? a.C * WAVE : 0));  // [wn][64] offset of S[u]

    const uint32_t lane = lane_id();
    const uint32_t g = blockIdx.x * WAVE + lane;
    const bool valid = g < a.n_reads;
    const uint32_t r = valid ? (a.order ? a.order[g] : g) : 0;

    if (MODE == MODE_FUSED)
        for (uint32_t c = 0; c < a.C; ++c) { cnt[c * WAVE + lane] = 0; unq[c * WAVE + lane] = 0; }
    if (MODE == MODE_EMPLACE) rbase[lane] = valid ? a.read_bin[r] : 0;
    __syncthreads();

    const uint32_t k = a.k;
    const uint64_t INV5 = 0xCCCCCCCCCCCCCCCDULL;  // 5^-1 mod 2^64 (exact division of the reverse strand)
    uint32_t my_emitted = 0;                      // == ReadEntry::num_hashes_ (include/read_entry.hpp:89)
    uint32_t qhead = 0, qcount = 0;               // wave-uniform queue state
    // One probe round in flight per wavefront.  (Two rounds were tried: hipcc guards every consume and every 64-base chunk
    // boundary with s_waitcnt vmcnt(0), which drains ALL outstanding gathers, so a second round never stays in flight and
    // only costs ~90 VGPRs.  Counted vmcnt(N) waits would need every VMEM operation of the kernel in inline asm.)
    struct Pend {
        uint64_t w[5][W];
        uint32_t meta = 0;     // owner lane | per-read index << 6
        uint32_t n = 0;        // entries of the round (wave-uniform)
        bool has = false;      // per lane
        bool pending = false;  // wave-uniform
    };
    Pend p0;
    bool base_ready = true;  // wave-uniform: no untracked (asm) base prefetch is outstanding
    uint32_t consumed = 0;   // MODE_ROWS: log entries written so far (wave-uniform)
    const uint64_t wbase = (MODE == MODE_ROWS || MODE == MODE_LIST) ? a.wave_base[blockIdx.x] : 0;

    auto probe_consume = [&](Pend &P) {
        if (MODE == MODE_EMPLACE) { P.pending = false; return; }
        // the round's gathers are needed now; this full wait also completes the base prefetch issued before them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        base_ready = true;
        uint64_t acc[W];
#pragma unroll
        for (int w = 0; w < W; ++w) acc[w] = ~0ULL;
        if (P.has) {
            if (MODE == MODE_LIST) {
                acc[0] = P.w[0][0];
            } else {
#pragma unroll
                for (uint32_t i = 0; i < 5; ++i)
                    if (i < a.h) {
#pragma unroll
                        for (int w = 0; w < W; ++w) acc[w] &= P.w[i][w];
                    }
            }
        }
        const uint32_t owner = P.meta & 63u;
        if (MODE == MODE_FUSED) {
            uint64_t m = P.has ? (acc[0] & ((a.B >= 64) ? ~0ULL : ((1ULL << a.B) - 1))) : 0;
            const bool single = __popcll(m) == 1;
            while (m) {
                const uint32_t b = (uint32_t)__ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t c = (uint32_t)(a.b2c_packed >> (8 * b)) & 0xffu;
                atomicAdd(&cnt[c * WAVE + owner], 1u);
                if (single) atomicAdd(&unq[c * WAVE + owner], 1u);
            }
        } else {  // MODE_ROWS / MODE_LIST: append the round to the wavefront's log -- 64 consecutive entries per round, coalesced
            if (P.has) {
                const uint64_t e = wbase + consumed + lane;
                if (MODE == MODE_LIST) {
                    a.rows[e] = acc[0];
                    a.rowlog[e] = owner << 26;
                } else {
                    bool esc;
                    a.rowlog[e] = encode_row<W>(acc, a.B, owner, esc);
                    if (esc) {
                        uint64_t *dst = a.rows + e * W;
#pragma unroll
                        for (int w = 0; w < W; ++w) dst[w] = acc[w];
                    }
                }
            }
            consumed += P.n;
        }
        P.pending = false;
    };

    auto probe_issue = [&](Pend &P, uint32_t n_take) {
        __syncthreads();
        const bool has = lane < n_take;
        const uint32_t slot = (qhead + lane) & (QCAP - 1);
        const uint64_t val = qv[slot];
        const uint32_t meta = qm[slot];
        if (MODE == MODE_EMPLACE) {
            if (has) {
                const uint32_t bin = (uint32_t)rbase[meta & 63u];
                for (uint32_t i = 0; i < a.h; ++i) {
                    uint64_t row = hash_and_fit_row(val, c_ibf_seeds[i], a.S, a.shift);
                    if (row >= a.row_begin && row < a.row_end)
                        atomicOr((unsigned long long *)&a.words_rw[(row - a.row_begin) * W + (bin >> 6)], 1ULL << (bin & 63));
                }
            }
        } else {
            P.has = has;
            P.meta = meta;
            P.n = n_take;
            if (MODE == MODE_LIST) {
                if (has) P.w[0][0] = val;
            } else if (has) {
                uint64_t rows_[5];
#pragma unroll
                for (uint32_t i = 0; i < 5; ++i)
                    if (i < a.h) rows_[i] = hash_and_fit_row(val, c_ibf_seeds[i], a.S, a.shift);
#pragma unroll
                for (uint32_t i = 0; i < 5; ++i)
                    if (i < a.h) {
                        if (a.ablate & 3u) { P.w[i][0] = rows_[i]; for (int w = 1; w < W; ++w) P.w[i][w] = val; continue; }
                        const uint64_t *p = a.words + (rows_[i] - a.row_begin) * W;
                        // Cache policy by index size (tools/gather_policy_bench.hip): from a table far larger than the 256 MiB Infinity
                        // Cache a probed line is never reused and `nt` gathers sustain 11 % more (54.3 vs 48.8 G/s at 39 GB); from a
                        // ~1 GiB table a quarter of the probes hit the Infinity Cache and the default policy wins (53.2 vs 49.8 G/s).
                        if (a.nt_probes) {
#pragma unroll
                            for (int w = 0; w < W; ++w) P.w[i][w] = __builtin_nontemporal_load(p + w);
                        } else {
#pragma unroll
                            for (int w = 0; w < W; ++w) P.w[i][w] = p[w];
                        }
                    }
            }
            P.pending = true;
        }
        qhead = (qhead + n_take) & (QCAP - 1);
        qcount -= n_take;
        __syncthreads();
    };

    // one probe turn: retire the round in flight (issued ~11 bases ago), then issue the next one
    auto probe_turn = [&](uint32_t n_take) {
        if (p0.pending) probe_consume(p0);
        probe_issue(p0, n_take);
    };

    // per-segment rolling state (hoisted so that the step body below can be one generic lambda)
    uint64_t fwd = 0, rc = 0, hist2 = 0, mv = 0, pv = 0;
    uint32_t histn = 0, q = 0, pq = 0;
    uint32_t t = 0, blk = 0;  // wave-uniform: offset of the newest value in its block, start index of that block
    uint32_t L = 0;
    bool has_nmask = false;

    // One base step.  STEADY: i >= k + wn and every lane is active (compile-time tag: the start-up, first-window, short-read
    // and activity checks disappear).  HASN: the batch carries an N mask.
    auto step = [&](auto STEADY, auto HASN, const uint32_t i, const uint32_t j, const uint32_t cur, const uint32_t ncur) {
        constexpr bool kSteady = decltype(STEADY)::value, kHasN = decltype(HASN)::value;
        const bool act = kSteady || i < L;
        bool emit = false;
        const bool have_value = kSteady || i + 1 >= k;  // wave-uniform
        if (act) {
            const uint32_t code = (cur >> (j * 2)) & 3u;
            const uint32_t nf = (kHasN && has_nmask) ? ((ncur >> (i & 31u)) & 1u) : 0u;
            // dna5 ranks A0 C1 G2 N3 T4; complement table [4,2,1,3,0]
            const uint32_t d_in = nf ? 3u : code + (code == 3u);
            const uint32_t cd_in = nf ? 3u : (3u - code) + (code == 0u);
            uint32_t d_out = 0, cd_out = 0;
            if (kSteady || i >= k) {
                const uint32_t oc = (uint32_t)(hist2 >> (2 * (k - 1))) & 3u;
                const uint32_t on = kHasN ? ((histn >> (k - 1)) & 1u) : 0u;
                d_out = on ? 3u : oc + (oc == 3u);
                cd_out = on ? 3u : (3u - oc) + (oc == 0u);
            }
            hist2 = (hist2 << 2) | code;
            if (kHasN) histn = (histn << 1) | nf;
            fwd = (fwd - (uint64_t)d_out * a.powk1) * 5u + d_in;
            rc = (rc - cd_out) * INV5 + (uint64_t)cd_in * a.powk1;
            if (have_value) {
                const uint32_t p = i + 1 - k;  // index of this canonical value (== blk + t)
                const uint64_t vf = fwd ^ a.seed, vr = rc ^ a.seed;
                const uint64_t v = vf < vr ? vf : vr;
                // running rightmost minimum of the current block's prefix [blk, p]
                if (t == 0 || v <= pv) { pv = v; pq = p; }
                if (!kSteady && p < wn) {  // first window: its rightmost minimum is the prefix minimum of block 0
                    mv = pv; q = pq;
                    emit = (p == wn - 1);
                } else if (q + wn == p) {  // tracked minimum left the window: rightmost minimum of the new one
                    mv = pv; q = pq;
                    if (t + 1 < wn) {
                        const uint64_t sv = ring[(t + 1) * WAVE + lane];
                        if (sv < pv) { mv = sv; q = blk - wn + spos[(t + 1) * WAVE + lane]; }
                    }
                    emit = true;
                } else if (v < mv) {
                    mv = v; q = p; emit = true;
                }
                ring[t * WAVE + lane] = v;
                // sequence shorter than one window: a single minimiser over all its values
                if (!kSteady && i + 1 == L && p + 1 < wn) emit = true;
            }
        }
        if (have_value) {
            if (t + 1 == wn) {
                // block end: turn the raw values of this block into rightmost suffix minima, in place
                uint64_t sv = ring[(wn - 1) * WAVE + lane];
                uint32_t sp = wn - 1;
                spos[(wn - 1) * WAVE + lane] = (uint8_t)sp;
#pragma unroll
                for (int u = (int)wn - 2; u >= 1; --u) {
                    const uint64_t x = ring[u * WAVE + lane];
                    if (x < sv) { sv = x; sp = (uint32_t)u; }
                    ring[u * WAVE + lane] = sv;
                    spos[u * WAVE + lane] = (uint8_t)sp;
                }
                t = 0; blk += wn;
            } else {
                ++t;
            }
        }
        const uint64_t mask = __ballot(emit);
        if (mask) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (emit) {
                const uint32_t pos = (qhead + qcount + rank) & (QCAP - 1);
                qv[pos] = mv;
                qm[pos] = lane | (my_emitted << 6);
                ++my_emitted;
            }
            qcount += (uint32_t)__popcll(mask);
            if (qcount >= WAVE) probe_turn(WAVE);
        }
    };

    for (uint32_t s = 0; s < a.nseg; ++s) {
        L = valid ? (s == 0 ? a.len1[r] : a.len2[r]) : 0;
        const uint64_t off = valid ? (s == 0 ? a.off1[r] : a.off2[r]) : 0;
        const uint32_t maxL = wave_max_u32(L);
        const uint32_t minL = ~wave_max_u32(~L);  // lanes past the end of the batch have L = 0 and force the generic body
        const uint4 *bp = reinterpret_cast<const uint4 *>(a.bases + (off >> 4));   // 64 bases per uint4
        const uint2 *np = a.nmask ? reinterpret_cast<const uint2 *>(a.nmask + (off >> 5)) : nullptr;
        has_nmask = np != nullptr;
        fwd = 0; rc = 0; hist2 = 0; mv = 0; pv = 0; histn = 0; q = 0; pq = 0; t = 0; blk = 0;

        const uint32_t nchunk = (maxL + 63) >> 6;
        // Bases are prefetched one 64-base chunk ahead through asm loads.  hipcc would guard their use with s_waitcnt
        // vmcnt(0) at every chunk boundary and thereby drain the probe round in flight; instead the kernel knows that any
        // probe consume since the prefetch has already waited for it (base_ready) and only waits when none happened.
        u32x4_t wnext = {0u, 0u, 0u, 0u};
        u32x2_t nnext = {0u, 0u};
        if (L > 0) { wnext = asm_load_b128(bp); if (np) nnext = asm_load_b64(np); }
        base_ready = false;
        for (uint32_t c = 0; c < nchunk; ++c) {
            if (!base_ready || (a.ablate & 4u)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); base_ready = true; }  // bit 2: A/B diagnostic
            asm volatile("" : "+v"(wnext), "+v"(nnext));  // the values are read only after the wait above
            const uint4 wcur = make_uint4(wnext.x, wnext.y, wnext.z, wnext.w);
            const uint2 ncur2 = make_uint2(nnext.x, nnext.y);
            if ((c + 1) * 64 < maxL) {
                if ((c + 1) * 64 < L) { wnext = asm_load_b128(bp + c + 1); if (np) nnext = asm_load_b64(np + c + 1); }
                base_ready = false;
            }
            for (uint32_t dd = 0; dd < 4; ++dd) {
                const uint32_t cur = dd == 0 ? wcur.x : dd == 1 ? wcur.y : dd == 2 ? wcur.z : wcur.w;
                const uint32_t ncur = dd < 2 ? ncur2.x : ncur2.y;
                const uint32_t ibase = c * 64 + dd * 16;
                if (ibase >= maxL) break;
                const uint32_t jn = maxL - ibase < 16 ? maxL - ibase : 16;
                // Steady-state dword: every lane still has all 16 bases, and all of them lie past the warm-up and the first
                // window, so the specialised body drops the activity predicate and the start-up / short-read cases.
                const bool steady = ibase >= k + wn && ibase + 16 <= minL;
                if (steady && !np) {
                    for (uint32_t j = 0; j < 16; ++j) step(std::true_type(), std::false_type(), ibase + j, j, cur, ncur);
                } else if (steady) {
                    for (uint32_t j = 0; j < 16; ++j) step(std::true_type(), std::true_type(), ibase + j, j, cur, ncur);
                } else {
                    for (uint32_t j = 0; j < jn; ++j) step(std::false_type(), std::true_type(), ibase + j, j, cur, ncur);
                }
            }
        }
    }
    if (qcount) probe_turn(qcount);
    if (p0.pending) probe_consume(p0);

    __syncthreads();
    if ((MODE == MODE_ROWS || MODE == MODE_LIST) && lane == 0) a.wave_count[blockIdx.x] = consumed;
    if (valid && MODE != MODE_EMPLACE) {
        a.num_hashes[r] = my_emitted;
        if (MODE == MODE_FUSED) {
            for (uint32_t c = 0; c < a.C; ++c) {
                a.counts[(size_t)r * a.C + c] = cnt[c * WAVE + lane];
                a.unique[(size_t)r * a.C + c] = unq[c * WAVE + lane];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_count_wavelog: ReadEntry::get_counts (include/read_entry.hpp:92-138) on the row log of one wavefront of
// k_minimise_probe (64 reads).  One workgroup per log; all per-read state lives in LDS.
// ------------------------------------------------------------------------------------------------
struct K2Args {
    const uint64_t *rows;
    const uint32_t *rowlog;
    const uint64_t *wave_base;
    const uint32_t *wave_count;
    const uint32_t *order;
    uint32_t *counts, *unique;
    uint32_t n_reads, B, C, W;
    uint8_t b2c[256];
};

#define K2_THREADS 512
#define K2_UNROLL 4
template <int W>
__global__ __launch_bounds__(K2_THREADS) void k_count_wavelog(const K2Args a) {
    extern __shared__ __align__(16) unsigned char smem2[];
    const uint32_t B = a.B, C = a.C;
    uint32_t *tot = reinterpret_cast<uint32_t *>(smem2);                 // [64][B] total_bits_per_bin (:96-99)
    uint32_t *unq = tot + (size_t)WAVE * B;                              // [64][C]
    uint64_t *cmask = reinterpret_cast<uint64_t *>(unq + (size_t)WAVE * C + ((WAVE * C) & 1u));  // [64][W] chosen-bin mask
    uint8_t *chosen = reinterpret_cast<uint8_t *>(cmask + WAVE * W);     // [64][C]
    __shared__ uint8_t s_b2c[256];
    const uint32_t tid = threadIdx.x, g = blockIdx.x;
    for (uint32_t i = tid; i < WAVE * B; i += K2_THREADS) tot[i] = 0;
    for (uint32_t i = tid; i < WAVE * C; i += K2_THREADS) { unq[i] = 0; chosen[i] = 255; }
    for (uint32_t i = tid; i < WAVE * W; i += K2_THREADS) cmask[i] = 0;
    if (tid < 256) s_b2c[tid] = a.b2c[tid];
    __syncthreads();
    const uint32_t count = a.wave_count[g];
    const uint64_t base = a.wave_base[g];
    const uint64_t *rows = a.rows + base * W;
    const uint32_t *log = a.rowlog + base;
    const uint64_t lastmask = (B & 63u) ? ((1ULL << (B & 63u)) - 1) : ~0ULL;
    // pass A: per-bin totals of each of the 64 reads.  Four compact entries per load and K2_UNROLL loads in flight per thread:
    // the loop is latency-bound per thread, not bandwidth-bound.
    const uint4 *log4 = reinterpret_cast<const uint4 *>(log);
    const uint32_t count4 = (count + 3) / 4;
    auto pass_a = [&](const uint4 v4, uint32_t i4) {
        const uint32_t xs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t e = i4 * 4 + q;
            if (e >= count) break;
            const uint32_t x = xs[q], o = x >> 26, pc = x & 3u;
            if (pc) {
                atomicAdd(&tot[o * B + ((x >> 2) & 0xffu)], 1u);
                if (pc > 1) atomicAdd(&tot[o * B + ((x >> 10) & 0xffu)], 1u);
                if (pc > 2) atomicAdd(&tot[o * B + ((x >> 18) & 0xffu)], 1u);
            } else if (x & 0x3fcu) {  // escaped: more than three set bins, full row in the side buffer
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    uint64_t r = rows[(size_t)e * W + w];
                    if (w == W - 1) r &= lastmask;
                    while (r) {
                        const uint32_t b = (uint32_t)__ffsll((long long)r) - 1;
                        r &= r - 1;
                        atomicAdd(&tot[o * B + w * 64 + b], 1u);
                    }
                }
            }
        }
    };
    for (uint32_t i4 = tid; i4 < count4; i4 += K2_THREADS * K2_UNROLL) {
        uint4 v[K2_UNROLL];
#pragma unroll
        for (int u = 0; u < K2_UNROLL; ++u) {
            const uint32_t j = i4 + u * K2_THREADS;
            v[u] = j < count4 ? log4[j] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < K2_UNROLL; ++u) pass_a(v[u], i4 + u * K2_THREADS);
    }
    __syncthreads();
    // first bin with the strictly largest total per category (:102-115): one thread per (read, category), bins scanned in
    // ascending order with the running best in registers
    for (uint32_t i = tid; i < WAVE * C; i += K2_THREADS) {
        const uint32_t o = i / C, c = i % C;
        uint32_t best = 255, bestv = 0;
        for (uint32_t b = 0; b < B; ++b) {
            const uint32_t t = tot[o * B + b];
            if (s_b2c[b] == c && (best == 255 || t > bestv)) { best = b; bestv = t; }
        }
        chosen[i] = (uint8_t)best;
        if (best != 255) atomicOr((unsigned long long *)&cmask[o * W + (best >> 6)], 1ULL << (best & 63u));
        const uint32_t gi = g * WAVE + o;
        if (gi < a.n_reads) a.counts[(size_t)a.order[gi] * C + c] = best == 255 ? 0u : bestv;  // a category without bins keeps 0
    }
    __syncthreads();
    // pass B: a minimiser is a unique hit if exactly one category's chosen bin contains it (:121-136)
    auto pass_b = [&](const uint4 v4, uint32_t i4) {
        const uint32_t xs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t e = i4 * 4 + q;
            if (e >= count) break;
            const uint32_t x = xs[q], o = x >> 26, pc = x & 3u;
            uint32_t found = 0, fbin = 0;
            if (pc) {
                for (uint32_t j = 0; j < pc; ++j) {
                    const uint32_t b = (x >> (2 + 8 * j)) & 0xffu;
                    if ((cmask[o * W + (b >> 6)] >> (b & 63u)) & 1ULL) { ++found; fbin = b; }
                }
            } else if (x & 0x3fcu) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const uint64_t r = rows[(size_t)e * W + w] & cmask[o * W + w];
                    found += (uint32_t)__popcll(r);
                    if (r) fbin = w * 64 + (uint32_t)__ffsll((long long)r) - 1;
                }
            }
            if (found == 1) atomicAdd(&unq[o * C + s_b2c[fbin]], 1u);
        }
    };
    for (uint32_t i4 = tid; i4 < count4; i4 += K2_THREADS * K2_UNROLL) {
        uint4 v[K2_UNROLL];
#pragma unroll
        for (int u = 0; u < K2_UNROLL; ++u) {
            const uint32_t j = i4 + u * K2_THREADS;
            v[u] = j < count4 ? log4[j] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < K2_UNROLL; ++u) pass_b(v[u], i4 + u * K2_THREADS);
    }
    __syncthreads();
    for (uint32_t i = tid; i < WAVE * C; i += K2_THREADS) {
        const uint32_t o = i / C, c = i % C, gi = g * WAVE + o;
        if (gi < a.n_reads) a.unique[(size_t)a.order[gi] * C + c] = unq[i];
    }
}
static size_t k2_lds_bytes(uint32_t B, uint32_t C, uint32_t W) {
    return (size_t)WAVE * B * 4 + ((size_t)WAVE * C + ((WAVE * C) & 1u)) * 4 + (size_t)WAVE * W * 8 + (size_t)WAVE * C + 16;
}

// capacity of a wavefront's row log = total bases of its 64 reads (at most one emission per base), then an exclusive scan
__global__ void k_wave_caps(const uint32_t *order, const uint32_t *len1, const uint32_t *len2, uint32_t n, uint64_t *caps) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = 0;
    if (i < n) { const uint32_t r = order[i]; v = len1[r] + (len2 ? len2[r] : 0u); }
    uint64_t sum = v;
    for (int o = 32; o > 0; o >>= 1) sum += (uint64_t)__shfl_xor((long long)sum, o);
    if (lane_id() == 0 && i < n) caps[i / WAVE] = (sum + 3) & ~3ULL;  // log bases stay 16-byte aligned for uint4 reads of compact entries
}
__global__ __launch_bounds__(1024) void k_scan_u64(uint64_t *v, uint32_t n) {  // in place exclusive scan, single workgroup
    __shared__ uint64_t part[1024];
    const uint32_t per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = min(n, lo + per);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t run = 0; for (int i = 0; i < 1024; ++i) { const uint64_t t = part[i]; part[i] = run; run += t; } }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) { const uint64_t t = v[i]; v[i] = run; run += t; }
}

// ------------------------------------------------------------------------------------------------
// row-sharded ("hash-bin" sharded) probing: every rank holds rows [row_begin, row_end) of the IBF.  A probe word is
// written by exactly one rank (all others write 0), so one sum all-reduce of the partial buffer reconstructs it
// exactly; the AND over the h hash functions happens afterwards.
// ------------------------------------------------------------------------------------------------
struct ShardArgs {
    const uint64_t *list;        // minimiser values, wavefront log layout
    const uint64_t *wave_base;   // start of wavefront g's region in `list` / `rows`
    const uint64_t *cbase;       // start of wavefront g's entries in the COMPACT partial buffer
    const uint32_t *wave_count;
    const uint64_t *words;
    uint64_t S, row_begin, row_end;
    uint32_t shift, h, W;
    uint64_t *partial;           // [entry][h][W]
    uint64_t *rows;              // full-row side buffer of the wavefront log (k_and_partial, escaped rows only)
    uint32_t *rowlog;            // compact row log (owner in bits 26-31 on entry to k_and_partial)
    uint32_t B;
};
template <int W>
__global__ __launch_bounds__(256) void k_probe_partial(const ShardArgs a) {
    const uint32_t g = blockIdx.x, count = a.wave_count[g];
    const uint64_t *list = a.list + a.wave_base[g];
    uint64_t *out = a.partial + a.cbase[g] * a.h * W;
    for (uint32_t e = threadIdx.x; e < count; e += blockDim.x) {
        const uint64_t val = list[e];
        for (uint32_t i = 0; i < a.h; ++i) {
            const uint64_t row = hash_and_fit_row(val, c_ibf_seeds[i], a.S, a.shift);
            const bool mine = row >= a.row_begin && row < a.row_end;
            const uint64_t *p = a.words + (mine ? (row - a.row_begin) * W : 0);
#pragma unroll
            for (int w = 0; w < W; ++w) out[((size_t)e * a.h + i) * W + w] = mine ? __builtin_nontemporal_load(p + w) : 0ULL;
        }
    }
}
template <int W>
__global__ __launch_bounds__(256) void k_and_partial(const ShardArgs a) {
    const uint32_t g = blockIdx.x, count = a.wave_count[g];
    const uint64_t *in = a.partial + a.cbase[g] * a.h * W;
    uint64_t *rows = a.rows + a.wave_base[g] * W;
    uint32_t *log = a.rowlog + a.wave_base[g];
    for (uint32_t e = threadIdx.x; e < count; e += blockDim.x) {
        uint64_t acc[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            acc[w] = ~0ULL;
            for (uint32_t i = 0; i < a.h; ++i) acc[w] &= in[((size_t)e * a.h + i) * W + w];
        }
        bool esc;
        log[e] = encode_row<W>(acc, a.B, log[e] >> 26, esc);
        if (esc) {
#pragma unroll
            for (int w = 0; w < W; ++w) rows[(size_t)e * W + w] = acc[w];
        }
    }
}
__global__ __launch_bounds__(256) void k_compact_list(const ShardArgs a, uint64_t *out) {
    const uint32_t g = blockIdx.x, count = a.wave_count[g];
    const uint64_t *list = a.list + a.wave_base[g];
    uint64_t *dst = out + a.cbase[g];
    for (uint32_t e = threadIdx.x; e < count; e += blockDim.x) dst[e] = list[e];
}
__global__ void k_emplace_values(uint64_t *words, const uint64_t *values, uint64_t n, uint32_t bin, uint64_t S, uint64_t row_begin, uint64_t row_end,
                                 uint32_t shift, uint32_t h, uint32_t W) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = values[i];
        for (uint32_t j = 0; j < h; ++j) {
            const uint64_t row = hash_and_fit_row(v, c_ibf_seeds[j], S, shift);
            if (row >= row_begin && row < row_end) atomicOr((unsigned long long *)&words[(row - row_begin) * W + (bin >> 6)], 1ULL << (bin & 63));
        }
    }
}
__global__ void k_counts_to_u64(const uint32_t *c, uint32_t n, uint64_t *out /* n + 1 */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = c[i];
    else if (i == n) out[i] = 0;
}

// ------------------------------------------------------------------------------------------------
// k_model_call: apply_model + call_host / call_category
// ------------------------------------------------------------------------------------------------
struct K3Args {
    const uint32_t *num_hashes, *counts, *unique;
    const uint32_t *len1, *len2;
    const float *mean_quality, *compression;
    double *prob;
    uint8_t *call, *conf, *flags;
    const float *data;            // all KDE datasets back to back
    const uint32_t *tab;          // [4][C]: pos_off, pos_n, neg_off, neg_n per category
    ulonglong2 *memo;             // direct-mapped cache (category, num_hashes, unique) -> probability; may be null
    uint32_t memo_mask;
    uint32_t n_reads, C;
    float h_pos, h_neg, log_rate, rate;
    float min_quality, min_compression, cpt, lo_thr, min_pd, min_prd;
    uint32_t min_length;
    int32_t conf_thr;
    uint32_t min_hits, paired, host_index;
};

__device__ __forceinline__ uint32_t rl_u32(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ float rl_f32(float v, uint32_t lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)lane)); }

// KDEParams::prob / K (include/classify_stats.hpp:242-252): K evaluated in double, accumulated in float IN DATA ORDER.
// Evaluated by the whole wavefront for one wave-uniform x: lane j computes the term of point base + j (the expensive double
// exp), then the terms are added to the (uniform) float accumulator in ascending point order through readlane, so the sum
// is bit-identical to the reference's sequential loop while the exps run 64 wide.
// A term with |t| > 15 is exp(-112.5)/sqrt(2 pi) < 6e-50, which rounds to +0.0f, and adding +0.0f leaves the float
// accumulator unchanged -- skipping it is exact (it removes most of the narrow-bandwidth terms).  NaN compares false.
__device__ __forceinline__ float kde_prob_wave(const float *data, uint32_t n, float h, float x) {
    const uint32_t lane = __lane_id();
    float total = 0.0f;
    for (uint32_t base = 0; base < n; base += WAVE) {
        const uint32_t i = base + lane;
        float term = 0.0f;
        bool live = false;
        if (i < n) {
            const float t = (x - data[i]) / h;
            if (!(fabsf(t) > 15.0f)) {
                const double kd = exp(-((double)t * (double)t) / 2.0) / sqrt(2 * 3.141592653589793238463);
                term = (float)kd;
                live = true;
            }
        }
        uint64_t m = __ballot(live);
        while (m) {
            const uint32_t j = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            total += rl_f32(term, j);
        }
    }
    return total / (h * (float)n);
}

// Model::prob (include/classify_stats.hpp:370-389) of category c for a read with `uq` unique hits out of `nh` minimisers;
// c, uq, nh wave-uniform, every lane returns the same value
__device__ __forceinline__ double model_prob_wave(const K3Args &a, uint32_t c, uint32_t uq, uint32_t nh) {
    const float x = (float)uq / (float)nh;  // unique proportion (include/read_entry.hpp:140-150)
    float p_err;
    if (x != x) p_err = x;
    else if (x < 0.0f) p_err = 0.0f;
    else p_err = (float)exp((double)(a.log_rate - a.rate * x));  // stats::dexp(x, 300) = exp(log(300) - 300 x)
    float p_pos = kde_prob_wave(a.data + a.tab[c], a.tab[a.C + c], a.h_pos, x);
    const float p_neg = kde_prob_wave(a.data + a.tab[2 * a.C + c], a.tab[3 * a.C + c], a.h_neg, x);
    if (x == 1.0f) p_pos = 1.0f;
    const float total = p_err + p_pos + p_neg;
    return (double)(p_pos / total);  // probabilities_ starts at 1 and is multiplied once (:56,277)
}

// The probability is a pure function of (c, uq, nh) for a fixed model, and a batch holds few distinct triples (reads of
// similar length), so it is memoised in a persistent direct-mapped table.  An entry is {check, prob} with
// check = ~(key ^ bits(prob)): a torn, stale or empty (all-zero) entry fails the check and is simply recomputed, so no
// ordering between writers and readers is needed; racing writers store identical values.  Misses are resolved by the
// whole wavefront, one distinct (uq, nh) at a time (lanes that miss on the same pair share the evaluation).
__device__ __forceinline__ double model_prob_lookup(const K3Args &a, bool valid, uint32_t c, uint32_t uq, uint32_t nh) {
    const bool keyed = a.memo && nh < (1u << 28) && uq < (1u << 28);
    const uint64_t key = ((uint64_t)c << 56) | ((uint64_t)nh << 28) | uq;
    ulonglong2 *slot = a.memo + (mix64(key) & a.memo_mask);
    double p = 0.0;
    bool have = !valid;
    if (valid && keyed) {
        const ulonglong2 e = *slot;
        if (e.x == ~(key ^ e.y)) { p = __longlong_as_double((long long)e.y); have = true; }
    }
    uint64_t miss = __ballot(!have);
    const uint32_t lane = __lane_id();
    while (miss) {
        const uint32_t l = (uint32_t)__builtin_ctzll(miss);
        const uint32_t kuq = rl_u32(uq, l), knh = rl_u32(nh, l);
        const double v = model_prob_wave(a, c, kuq, knh);
        if (!have && uq == kuq && nh == knh) { p = v; have = true; }
        if (lane == l && keyed) {
            const uint64_t bits = (uint64_t)__double_as_longlong(v);
            *slot = make_ulonglong2(~(key ^ bits), bits);
        }
        miss = __ballot(!have);
    }
    return p;
}

__global__ __launch_bounds__(256) void k_model_call(const K3Args a) {
    const uint32_t r0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = r0 < a.n_reads;
    const uint32_t r = valid ? r0 : 0u;  // lanes past the end stay in the wave-wide miss handling, with nothing to ask
    const uint32_t C = a.C;
    const uint32_t nh = a.num_hashes[r];
    const uint32_t *cnts = a.counts + (size_t)r * C, *uq = a.unique + (size_t)r * C;
    double *prob = a.prob + (size_t)r * C;
    for (uint32_t c = 0; c < C; ++c) {
        const double p = model_prob_lookup(a, valid, c, uq[c], nh);
        if (valid) prob[c] = p;
    }
    if (!valid) return;
    const float mq = a.mean_quality ? a.mean_quality[r] : 0.0f;
    const float comp = a.compression ? a.compression[r] : 0.0f;
    const uint32_t length = a.len1[r] + (a.len2 ? a.len2[r] : 0u);
    uint8_t call = 255, flag = 0;
    uint32_t conf;
    if (!a.paired) {
        // call_host (include/read_entry.hpp:218-269)
        const uint32_t host = a.host_index, other = 1u - host;
        const double hu = (float)uq[host] / (float)nh, ou = (float)uq[other] / (float)nh, hp = prob[host], op = prob[other];
        uint32_t first = host, second = other;
        if (hu < ou) { first = other; second = host; }
        const uint32_t raw = uq[first] - uq[second];
        conf = raw > 255u ? 255u : raw;
        const bool gate = !((int32_t)conf < a.conf_thr) && !(mq < a.min_quality) && !(length < a.min_length) &&
                          !(comp < a.min_compression);
        if (gate) {
            const double dc = (double)conf;
            if (hu > ou && hu - ou > a.min_pd && hp > op && hp - op > a.min_prd && fmax(hp * dc, dc) >= a.cpt)
                call = (uint8_t)host;
            else if (hu < a.lo_thr && hu < ou && ou - hu > a.min_pd && hp < op && op - hp > a.min_prd &&
                     fmax(op * dc, dc) >= a.cpt)
                call = (uint8_t)other;
            const double scale = fmax(fabs(hp), fabs(op));
            if (fabs(fabs(hp - op) - (double)a.min_prd) <= 2e-6 * fmax(scale, 1e-300)) flag = 1;
            if (a.cpt > 0.0f && (fabs(fmax(hp * dc, dc) - a.cpt) <= 2e-6 * fmax(a.cpt, 1.0f) ||
                                 fabs(fmax(op * dc, dc) - a.cpt) <= 2e-6 * fmax(a.cpt, 1.0f))) flag = 1;
        }
    } else {
        // call_category (include/read_entry.hpp:157-216)
        uint32_t first = 0, second = 1;
        if (uq[second] > uq[first]) { first = 1; second = 0; }
        for (uint32_t i = 2; i < C; ++i)
            if (uq[i] > uq[second]) {
                second = i;
                if (uq[second] > uq[first]) { const uint32_t t = first; first = second; second = t; }
            }
        const uint32_t raw = uq[first] - uq[second];
        conf = raw > 255u ? 255u : raw;
        const bool gate = !(mq < a.min_quality) && !(length < a.min_length) && !(comp < a.min_compression);
        if (gate) {
            const double pf = prob[first], ps = prob[second];
            const float propf = (float)cnts[first] / (float)nh, props = (float)cnts[second] / (float)nh;
            if (ps == 0 && pf > 0) call = (uint8_t)first;
            else if ((int32_t)conf > a.conf_thr && pf > ps) call = (uint8_t)first;
            if (cnts[second] > cnts[first] || cnts[first] - cnts[second] < a.min_hits) call = 255;
            if (props > propf || propf - props < a.min_pd) call = 255;
            const double scale = fmax(fabs(pf), fabs(ps));
            if (fabs(pf - ps) <= 2e-6 * fmax(scale, 1e-300)) flag = 1;
        }
    }
    a.call[r] = call;
    a.conf[r] = (uint8_t)conf;
    a.flags[r] = flag;
}

// ------------------------------------------------------------------------------------------------
// length-class ordering (counting sort on an 8-steps-per-octave bucket of the total read length)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t len_bucket(uint32_t len) {
    if (len < 8) return len;
    const uint32_t e = 31u - (uint32_t)__clz((int)len);
    return (e - 2) * 8 + ((len >> (e - 3)) & 7u);  // 8..247
}
// Stable (deterministic) counting sort: the read -> wavefront assignment must be identical on every rank of the
// row-sharded mode, and reproducible run to run.  LEN_BLOCKS workgroups each own a contiguous chunk of reads.
#define LEN_BLOCKS 256
__global__ __launch_bounds__(256) void k_len_hist(const uint32_t *len1, const uint32_t *len2, uint32_t n, uint32_t *blockhist /* [LEN_BLOCKS][256] */) {
    __shared__ uint32_t sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t chunk = (n + LEN_BLOCKS - 1) / LEN_BLOCKS, lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&sh[len_bucket(len1[i] + (len2 ? len2[i] : 0u))], 1u);
    __syncthreads();
    blockhist[blockIdx.x * 256 + threadIdx.x] = sh[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_len_scan(uint32_t *blockhist /* in: counts; out: start position of (block, bucket) */) {
    __shared__ uint32_t tot[256], start[256];
    const uint32_t b = threadIdx.x;
    uint32_t t = 0;
    for (uint32_t k = 0; k < LEN_BLOCKS; ++k) t += blockhist[k * 256 + b];
    tot[b] = t;
    __syncthreads();
    if (b == 0) { uint32_t run = 0; for (int x = 255; x >= 0; --x) { start[x] = run; run += tot[x]; } }  // longest bucket first
    __syncthreads();
    uint32_t run = start[b];
    for (uint32_t k = 0; k < LEN_BLOCKS; ++k) { const uint32_t c = blockhist[k * 256 + b]; blockhist[k * 256 + b] = run; run += c; }
}
__global__ __launch_bounds__(256) void k_len_scatter(const uint32_t *len1, const uint32_t *len2, uint32_t n, const uint32_t *blockstart, uint32_t *order) {
    __shared__ uint32_t cursor[256];
    __shared__ uint16_t tile[256];
    cursor[threadIdx.x] = blockstart[blockIdx.x * 256 + threadIdx.x];
    const uint32_t chunk = (n + LEN_BLOCKS - 1) / LEN_BLOCKS, lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    for (uint32_t base = lo; base < hi; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t bk = i < hi ? len_bucket(len1[i] + (len2 ? len2[i] : 0u)) : 0xffffu;
        __syncthreads();
        tile[threadIdx.x] = (uint16_t)bk;
        __syncthreads();
        uint32_t rank = 0, same = 0;
        // uniform-length batches put the whole tile into one bucket: rank = thread index, no counting needed
        const bool uniform_tile = __syncthreads_and(tile[threadIdx.x] == tile[0] && bk != 0xffffu) != 0;
        if (uniform_tile) {
            rank = threadIdx.x; same = 256;
            order[cursor[bk] + rank] = i;
        } else if (i < hi) {
            for (uint32_t j = 0; j < 256; ++j) { const bool eq = tile[j] == bk; same += eq; rank += eq && j < threadIdx.x; }
            order[cursor[bk] + rank] = i;
        }
        __syncthreads();
        if (i < hi && rank == same - 1) cursor[bk] += same;  // the last read of each bucket in this tile advances the cursor
    }
}

// algorithmic bytes of a batch: sum ceil(L/4) + M*h*W*8 + (8 + 8C)
__global__ void k_batch_bytes(const uint32_t *len1, const uint32_t *len2, const uint32_t *num_hashes, uint32_t n,
                              uint32_t hW8, uint32_t outb, unsigned long long *acc /* [2]: bytes, minimisers */) {
    unsigned long long b = 0, m = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t L = len1[i] + (len2 ? len2[i] : 0u);
        b += (L + 3) / 4 + (unsigned long long)num_hashes[i] * hW8 + outb;
        m += num_hashes[i];
    }
    for (int o = 32; o > 0; o >>= 1) { b += __shfl_xor((long long)b, o); m += __shfl_xor((long long)m, o); }
    if (lane_id() == 0) { atomicAdd(&acc[0], b); atomicAdd(&acc[1], m); }
}

// ------------------------------------------------------------------------------------------------
// Elias-Fano decode on the device (loader)
// ------------------------------------------------------------------------------------------------
struct EfArgs {
    const uint64_t *high, *low;
    uint64_t n_high_words, high_bit0, ones_before, low_elem0, m_size, row_begin, row_end;
    uint64_t *words;
    unsigned long long *bad;
    uint32_t wl, W, B, TB;
};
__global__ __launch_bounds__(256) void k_ef_block_counts(const uint64_t *high, uint64_t n, uint32_t *block_ones) {
    __shared__ uint32_t s[256];
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    s[threadIdx.x] = i < n ? (uint32_t)__popcll(high[i]) : 0u;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) block_ones[blockIdx.x] = s[0];
}
__global__ __launch_bounds__(256) void k_ef_decode(const EfArgs a, const uint64_t *block_prefix /* exclusive, per 256-word block */) {
    __shared__ uint32_t s[256];
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    const uint64_t word = i < a.n_high_words ? a.high[i] : 0ULL;
    const uint32_t pc = (uint32_t)__popcll(word);
    s[threadIdx.x] = pc;
    __syncthreads();
    // exclusive scan of the 256 popcounts (Hillis-Steele)
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t v = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
        __syncthreads();
        s[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t k = a.ones_before + block_prefix[blockIdx.x] + (s[threadIdx.x] - pc);  // rank of this word's first one
    uint64_t x = word;
    const uint64_t bit0 = a.high_bit0 + i * 64;
    const uint64_t lowmask = a.wl >= 64 ? ~0ULL : ((1ULL << a.wl) - 1);
    while (x) {
        const uint32_t j = (uint32_t)__ffsll((long long)x) - 1;
        x &= x - 1;
        const uint64_t z = bit0 + j - k;  // zeros before this one = its high part
        uint64_t lo = 0;
        if (a.wl) {
            const uint64_t bit = (k - a.low_elem0) * a.wl, wd = bit >> 6, sh = bit & 63;
            lo = a.low[wd] >> sh;
            if (sh + a.wl > 64) lo |= a.low[wd + 1] << (64 - sh);
            lo &= lowmask;
        }
        const uint64_t pos = (a.wl >= 64 ? 0 : (z << a.wl)) | lo;
        ++k;
        if (pos >= a.m_size) { atomicAdd(a.bad, 1ULL); continue; }
        const uint64_t row = pos / a.TB;
        const uint32_t bin = (uint32_t)(pos % a.TB);
        if (bin >= a.B) { atomicAdd(a.bad, 1ULL); continue; }
        if (row < a.row_begin || row >= a.row_end) continue;
        atomicOr((unsigned long long *)&a.words[(row - a.row_begin) * a.W + (bin >> 6)], 1ULL << (bin & 63));
    }
}
__global__ __launch_bounds__(256) void k_bin_popcounts(const uint64_t *words, uint64_t n_rows, uint32_t W, unsigned long long *out /* [W*64] */) {
    __shared__ uint32_t s[256];
    s[threadIdx.x] = 0;
    __syncthreads();
    // thread = (word index within row, bit): W*64 <= 256 counters per block; rows strided over blocks
    const uint32_t tb = W * 64;
    for (uint64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        if (threadIdx.x < tb) s[threadIdx.x] += (uint32_t)((words[r * W + (threadIdx.x >> 6)] >> (threadIdx.x & 63)) & 1ULL);
        if ((r / gridDim.x) % 1048576 == 1048575) {  // spill before the 32-bit counters could overflow
            if (threadIdx.x < tb) { atomicAdd(&out[threadIdx.x], (unsigned long long)s[threadIdx.x]); s[threadIdx.x] = 0; }
        }
    }
    if (threadIdx.x < tb) atomicAdd(&out[threadIdx.x], (unsigned long long)s[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// synthetic workload kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_synth_genomes(uint32_t *out, uint64_t n_dwords, uint64_t seed) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_dwords; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = (uint32_t)mix64(seed ^ (i * 0xD1342543DE82EF95ULL));
}
__global__ void k_synth_fill(uint64_t *words, uint64_t n_rows, uint64_t row_begin, uint32_t W, uint32_t B, uint64_t seed, uint32_t thr16) {
    // every bit of the user bins is set with probability thr16 / 65536 (four 16-bit lotteries per mix64)
    const uint64_t n_words = n_rows * W;
    for (uint64_t li = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; li < n_words; li += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = row_begin * W + li;  // global word index: a shard holds exactly the words of the full index
        const uint32_t w = (uint32_t)(i % W);
        uint64_t x = 0;
        for (uint32_t j = 0; j < 16; ++j) {
            const uint64_t z = mix64(seed ^ (i * 16 + j) * 0x9E3779B97F4A7C15ULL);
            for (uint32_t t = 0; t < 4; ++t)
                if (((z >> (16 * t)) & 0xffffu) < thr16) x |= 1ULL << (j * 4 + t);
        }
        const uint32_t lo = w * 64;
        if (B < lo + 64) x &= (B <= lo) ? 0ULL : ((1ULL << (B - lo)) - 1);
        words[li] = x;
    }
}
struct SynthReadsArgs {
    const uint32_t *genomes;
    uint64_t n_genomes, genome_len, seed, n_reads, first_id;
    uint32_t len_min, len_max, sub_thr32, rand_thr32;
    uint32_t *bases;
    const uint64_t *off;
    const uint32_t *len;
};
__global__ void k_synth_read_layout(uint64_t seed, uint64_t first_id, uint64_t n_reads, uint32_t len_min, uint32_t len_max, uint32_t *len) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n_reads) {
        uint32_t L = len_min;
        if (len_max > len_min) {  // log-uniform in [len_min, len_max]
            const double u = (double)(mix64(seed ^ ((first_id + i) * 0xA24BAED4963EE407ULL)) >> 11) * (1.0 / 9007199254740992.0);
            L = (uint32_t)(exp(log((double)len_min) + u * (log((double)len_max) - log((double)len_min))));
            L = L < len_min ? len_min : (L > len_max ? len_max : L);
        }
        len[i] = L;
    }
}
__global__ void k_synth_offsets(const uint32_t *len, uint64_t n_reads, uint64_t *off, uint64_t *total) {
    // single-thread exclusive scan of padded lengths (synthetic fabrication only, not on the timed path)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t run = 0;
        for (uint64_t i = 0; i < n_reads; ++i) { off[i] = run; run += ((uint64_t)len[i] + 63) & ~63ULL; }
        *total = run;
    }
}
__global__ void k_synth_reads(const SynthReadsArgs a) {
    // one thread per output dword (16 bases) of one read; grid.y = read chunks
    const uint64_t read = blockIdx.y + (uint64_t)blockIdx.z * gridDim.y;
    if (read >= a.n_reads) return;
    const uint32_t L = a.len[read];
    const uint32_t nd = (L + 63) / 64 * 4;
    const uint64_t rs = mix64(a.seed ^ ((a.first_id + read) * 0x9FB21C651E98DF25ULL));
    const bool is_random = a.n_genomes == 0 || (uint32_t)(rs >> 32) < a.rand_thr32;
    const uint64_t g = a.n_genomes ? (mix64(rs + 1) % a.n_genomes) : 0;
    const uint64_t span = a.genome_len > L ? a.genome_len - L : 1;
    const uint64_t start = mix64(rs + 2) % span;
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < nd; d += gridDim.x * blockDim.x) {
        uint32_t out = 0;
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t pos = d * 16 + j;
            if (pos >= L) break;
            const uint64_t z = mix64(rs ^ ((uint64_t)pos * 0xC2B2AE3D27D4EB4FULL));
            uint32_t code;
            if (is_random) code = (uint32_t)z & 3u;
            else {
                const uint64_t gp = g * a.genome_len + start + pos;
                code = (a.genomes[gp >> 4] >> ((gp & 15) * 2)) & 3u;
                if ((uint32_t)(z >> 32) < a.sub_thr32) code = (code + 1 + (uint32_t)(z % 3)) & 3u;  // uniform other base
            }
            out |= code << (2 * j);
        }
        a.bases[(a.off[read] >> 4) + d] = out;
    }
}
__global__ void k_fill_f32(float *p, uint64_t n, float v) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_chunk_layout(uint64_t n_genomes, uint64_t genome_len, uint32_t chunk, uint32_t overlap, uint64_t chunks_per_genome,
                               uint64_t *off, uint32_t *len) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n_genomes * chunks_per_genome) return;
    const uint64_t g = i / chunks_per_genome, c = i % chunks_per_genome;
    const uint64_t start = c * chunk;  // multiple of 64
    const uint64_t end = (start + chunk + overlap < genome_len) ? start + chunk + overlap : genome_len;
    off[i] = g * genome_len + start;
    len[i] = (uint32_t)(end - start);
}

// ------------------------------------------------------------------------------------------------
// host objects
// ------------------------------------------------------------------------------------------------
struct chn_index {
    chn_index_desc d;
    uint64_t *words = nullptr;
    uint64_t rows_local = 0;
    bool single_bin_categories = false;  // every category owns exactly one bin
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return CHN_OK;
        if (p) { HIPCHK(hipFree(p)); p = nullptr; cap = 0; }
        HIPCHK(hipMalloc(&p, bytes));
        cap = bytes;
        return CHN_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

struct HostModel {  // deep copy of chn_model
    bool set = false;
    uint32_t C = 0;
    std::vector<std::vector<float>> pos, neg;
    float h_pos = 0.1f, h_neg = 0.001f, rate = 300.0f;
    float min_quality = 15.0f, min_compression = 0, cpt = 0, lo_thr = 0.05f, min_pd = 0.04f, min_prd = 0;
    uint32_t min_length = 140;
    int8_t conf_thr = 7;
    uint8_t min_hits = 0, paired = 0, host_index = 0;
};

// Per-batch state.  Two slots let batch i+1 be submitted while batch i's model+call kernel (fp64 VALU work on a side
// stream) is still running: that kernel then overlaps the HBM-bound minimise+probe kernel of the next batch.
struct Slot {
    DevBuf d_num_hashes, d_counts, d_unique, d_prob, d_call, d_conf, d_flags, d_acc;
    DevBuf d_len1, d_len2, d_mq, d_comp;  // staging of the small per-read arrays of host batches (read by k_model_call)
    DevBuf d_bases, d_nmask, d_off1, d_off2;  // staging of the large arrays: per slot, so batch i+1 uploads while batch i computes
    hipEvent_t uploaded = nullptr;
    const uint32_t *len1 = nullptr, *len2 = nullptr;
    const float *mq = nullptr, *comp = nullptr;
    uint64_t n_reads = 0;
    bool host_batch = false, model_ran = false;
    std::vector<uint32_t> h_len1, h_len2;
    std::vector<float> h_mq, h_comp;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // K1 0-1, K2 2-3, K3 4-5, chain 6-7
    hipEvent_t k2_done = nullptr, done = nullptr;
    bool ev_used[4] = {false, false, false, false};
};

struct chn_stream {
    chn_index *idx = nullptr;
    chn_stream_cfg cfg;
    hipStream_t stream = nullptr;   // uploads, ordering, minimise+probe, count
    hipStream_t stream2 = nullptr;  // model+call
    hipStream_t stream0 = nullptr;  // host -> device uploads of host batches (truly asynchronous when the caller's memory is pinned)
    DevBuf d_order, d_hist, d_rows, d_rowown, d_wbase, d_wcount, d_model;
    DevBuf d_memo;                    // k_model_call memo table (cleared whenever the model changes)
    DevBuf d_list, d_cbase;           // row-sharded mode: minimiser value log, compact entry offsets
    uint64_t shard_entries = 0;
    bool shard_open = false;
    Slot slot[2];
    int head = 0;      // slot of the next submit
    int inflight = 0;  // batches submitted and not yet waited for (FIFO)
    HostModel model;
    K3Args k3;
    // profiling
    double prof_ms[4] = {0, 0, 0, 0};
    uint64_t prof_n[4] = {0, 0, 0, 0};
    uint64_t last_bytes = 0, last_min = 0;
};

static uint64_t pow5(unsigned e) { uint64_t p = 1; while (e--) p *= 5; return p; }

extern "C" int chn_index_create(const chn_index_desc *desc, chn_index **out) {
    if (!desc || !out || desc->struct_size != sizeof(chn_index_desc)) return fail(CHN_E_INVALID, "chn_index_create: bad descriptor");
    const chn_index_desc &d = *desc;
    if (d.kmer_size < 1 || d.kmer_size > 27) return fail(CHN_E_INVALID, "kmer_size must be in 1..27 (5^k has to fit 64 bits)");
    if (d.window_size < d.kmer_size) return fail(CHN_E_INVALID, "window_size must be >= kmer_size");
    if (d.hash_funs < 1 || d.hash_funs > 5) return fail(CHN_E_INVALID, "hash_funs must be in 1..5");
    if (d.bins < 1 || d.bins > 255 || d.technical_bins != ((d.bins + 63) / 64) * 64 || d.bin_words != d.technical_bins / 64)
        return fail(CHN_E_INVALID, "IBF header inconsistent: technical_bins must be 64*ceil(bins/64), bin_words = technical_bins/64");
    if (d.bin_size == 0 || d.hash_shift != (uint64_t)__builtin_clzll(d.bin_size)) return fail(CHN_E_INVALID, "hash_shift must be countl_zero(bin_size)");
    if (d.num_categories < 1) return fail(CHN_E_INVALID, "num_categories must be >= 1");
    for (uint64_t b = 0; b < d.bins; ++b)
        if (d.bin_to_category[b] >= d.num_categories) return fail(CHN_E_INVALID, "bin_to_category entry out of range");
    chn_index *idx = new (std::nothrow) chn_index();
    if (!idx) return fail(CHN_E_NOMEM, "host allocation failed");
    idx->d = d;
    if (idx->d.row_begin == 0 && idx->d.row_end == 0) idx->d.row_end = d.bin_size;
    if (idx->d.row_end > d.bin_size || idx->d.row_begin >= idx->d.row_end) { delete idx; return fail(CHN_E_INVALID, "bad row shard"); }
    idx->rows_local = idx->d.row_end - idx->d.row_begin;
    std::vector<int> per_cat(d.num_categories, 0);
    for (uint64_t b = 0; b < d.bins; ++b) per_cat[d.bin_to_category[b]]++;
    idx->single_bin_categories = true;
    for (int c : per_cat) if (c != 1) idx->single_bin_categories = false;
    hipError_t e = hipSetDevice(d.device);
    if (e == hipSuccess) e = hipMalloc((void **)&idx->words, idx->rows_local * d.bin_words * 8);
    if (e != hipSuccess) { delete idx; return fail(e == hipErrorOutOfMemory ? CHN_E_NOMEM : CHN_E_HIP, std::string("index allocation: ") + hipGetErrorString(e)); }
    e = hipMemset(idx->words, 0, idx->rows_local * d.bin_words * 8);
    if (e != hipSuccess) { (void)hipFree(idx->words); delete idx; return fail(CHN_E_HIP, std::string("index memset: ") + hipGetErrorString(e)); }
    *out = idx;
    return CHN_OK;
}

extern "C" int chn_index_upload_rows(chn_index *idx, uint64_t row_begin, uint64_t n_rows, const uint64_t *host_words) {
    if (!idx || !host_words) return fail(CHN_E_INVALID, "chn_index_upload_rows: null argument");
    if (row_begin < idx->d.row_begin || row_begin + n_rows > idx->d.row_end) return fail(CHN_E_INVALID, "rows outside this shard");
    HIPCHK(hipSetDevice(idx->d.device));
    HIPCHK(hipMemcpy(idx->words + (row_begin - idx->d.row_begin) * idx->d.bin_words, host_words, n_rows * idx->d.bin_words * 8, hipMemcpyHostToDevice));
    return CHN_OK;
}
extern "C" int chn_index_download_rows(chn_index *idx, uint64_t row_begin, uint64_t n_rows, uint64_t *host_words) {
    if (!idx || !host_words) return fail(CHN_E_INVALID, "chn_index_download_rows: null argument");
    if (row_begin < idx->d.row_begin || row_begin + n_rows > idx->d.row_end) return fail(CHN_E_INVALID, "rows outside this shard");
    HIPCHK(hipSetDevice(idx->d.device));
    HIPCHK(hipMemcpy(host_words, idx->words + (row_begin - idx->d.row_begin) * idx->d.bin_words, n_rows * idx->d.bin_words * 8, hipMemcpyDeviceToHost));
    return CHN_OK;
}
extern "C" int chn_index_decode_ef(chn_index *idx, uint64_t m_size, uint32_t wl, const uint64_t *high, uint64_t high_bit0, uint64_t n_high_words,
                                   uint64_t ones_before, const uint64_t *low, uint64_t low_elem0, uint64_t n_low_words, uint64_t *bad_bits) {
    if (!idx || !high || (wl && !low) || !bad_bits) return fail(CHN_E_INVALID, "chn_index_decode_ef: null argument");
    if ((high_bit0 & 63) || wl > 64 || low_elem0 > ones_before) return fail(CHN_E_INVALID, "chn_index_decode_ef: bad slice description");
    const chn_index_desc &d = idx->d;
    if (m_size != d.technical_bins * d.bin_size) return fail(CHN_E_INVALID, "chn_index_decode_ef: m_size != technical_bins * bin_size");
    if (n_high_words == 0) return CHN_OK;
    if (n_high_words > (1ULL << 31)) return fail(CHN_E_INVALID, "chn_index_decode_ef: slice too large (at most 2^31 words)");
    HIPCHK(hipSetDevice(d.device));
    const uint64_t n_blocks = (n_high_words + 255) / 256;
    uint64_t *d_high = nullptr, *d_low = nullptr, *d_prefix = nullptr;
    uint32_t *d_counts = nullptr;
    unsigned long long *d_bad = nullptr;
    hipError_t e = hipMalloc((void **)&d_high, n_high_words * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_low, (n_low_words + 2) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_counts, n_blocks * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_prefix, n_blocks * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_bad, 8);
    if (e == hipSuccess) e = hipMemcpy(d_high, high, n_high_words * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_low, 0, (n_low_words + 2) * 8);
    if (e == hipSuccess && n_low_words) e = hipMemcpy(d_low, low, n_low_words * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_bad, 0, 8);
    std::vector<uint32_t> counts(n_blocks);
    std::vector<uint64_t> prefix(n_blocks);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_ef_block_counts, dim3((uint32_t)n_blocks), dim3(256), 0, 0, d_high, n_high_words, d_counts);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(counts.data(), d_counts, n_blocks * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        uint64_t run = 0;
        for (uint64_t b = 0; b < n_blocks; ++b) { prefix[b] = run; run += counts[b]; }
        e = hipMemcpy(d_prefix, prefix.data(), n_blocks * 8, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
        EfArgs a;
        a.high = d_high; a.low = d_low; a.n_high_words = n_high_words; a.high_bit0 = high_bit0; a.ones_before = ones_before; a.low_elem0 = low_elem0;
        a.m_size = m_size; a.row_begin = d.row_begin; a.row_end = d.row_end; a.words = idx->words; a.bad = d_bad; a.wl = wl;
        a.W = (uint32_t)d.bin_words; a.B = (uint32_t)d.bins; a.TB = (uint32_t)d.technical_bins;
        hipLaunchKernelGGL(k_ef_decode, dim3((uint32_t)n_blocks), dim3(256), 0, 0, a, d_prefix);
        e = hipGetLastError();
    }
    unsigned long long bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_high); (void)hipFree(d_low); (void)hipFree(d_counts); (void)hipFree(d_prefix); (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? CHN_E_NOMEM : CHN_E_HIP, std::string("chn_index_decode_ef: ") + hipGetErrorString(e));
    *bad_bits = bad;
    return CHN_OK;
}

extern "C" int chn_index_bin_popcounts(chn_index *idx, uint64_t *out) {
    if (!idx || !out) return fail(CHN_E_INVALID, "chn_index_bin_popcounts: null argument");
    const chn_index_desc &d = idx->d;
    HIPCHK(hipSetDevice(d.device));
    unsigned long long *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_out, d.technical_bins * 8));
    HIPCHK(hipMemset(d_out, 0, d.technical_bins * 8));
    hipLaunchKernelGGL(k_bin_popcounts, dim3(4096), dim3(256), 0, 0, idx->words, idx->rows_local, (uint32_t)d.bin_words, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d_out, d.technical_bins * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(CHN_E_HIP, std::string("chn_index_bin_popcounts: ") + hipGetErrorString(e));
    return CHN_OK;
}

extern "C" int chn_index_device_words(chn_index *idx, uint64_t **device_words, uint64_t *n_words) {
    if (!idx) return fail(CHN_E_INVALID, "null index");
    if (device_words) *device_words = idx->words;
    if (n_words) *n_words = idx->rows_local * idx->d.bin_words;
    return CHN_OK;
}
extern "C" int chn_index_get_desc(const chn_index *idx, chn_index_desc *out) {
    if (!idx || !out) return fail(CHN_E_INVALID, "null argument");
    *out = idx->d;
    return CHN_OK;
}
extern "C" int chn_index_destroy(chn_index *idx) {
    if (!idx) return CHN_OK;
    (void)hipSetDevice(idx->d.device);
    if (idx->words) (void)hipFree(idx->words);
    delete idx;
    return CHN_OK;
}

// ---- model --------------------------------------------------------------------------------------
static std::vector<float> g_def_pos, g_def_neg;
static const float *g_def_pos_ptr[CHN_MAX_CATEGORIES], *g_def_neg_ptr[CHN_MAX_CATEGORIES];
static uint32_t g_def_n_pos[CHN_MAX_CATEGORIES], g_def_n_neg[CHN_MAX_CATEGORIES];

extern "C" int chn_model_default(chn_model *m, uint32_t num_categories, uint8_t host_index, int paired) {
    if (!m || num_categories < 1 || num_categories > CHN_MAX_CATEGORIES) return fail(CHN_E_INVALID, "chn_model_default: bad argument");
    if (g_def_pos.empty()) {
        for (double v : CHN_DEFAULT_POS) g_def_pos.push_back((float)v);
        for (double v : CHN_DEFAULT_NEG) g_def_neg.push_back((float)v);
        std::sort(g_def_pos.begin(), g_def_pos.end());  // KDEParams constructor sorts (include/classify_stats.hpp:214-218)
        std::sort(g_def_neg.begin(), g_def_neg.end());
    }
    for (uint32_t c = 0; c < num_categories; ++c) {
        g_def_pos_ptr[c] = g_def_pos.data(); g_def_neg_ptr[c] = g_def_neg.data();
        g_def_n_pos[c] = (uint32_t)g_def_pos.size(); g_def_n_neg[c] = (uint32_t)g_def_neg.size();
    }
    std::memset(m, 0, sizeof(*m));
    m->struct_size = sizeof(chn_model);
    m->num_categories = num_categories;
    m->pos_data = g_def_pos_ptr; m->pos_n = g_def_n_pos; m->neg_data = g_def_neg_ptr; m->neg_n = g_def_n_neg;
    m->h_pos = 0.1f; m->h_neg = 0.001f; m->err_rate = 300.0f;
    m->min_quality = 15.0f; m->min_length = paired ? 80 : 140; m->min_compression = 0.0f;
    m->confidence_threshold = 7; m->min_hits = 0; m->paired = paired ? 1 : 0; m->host_index = host_index;
    m->confidence_probability_threshold = 0.0f; m->host_unique_prop_lo_threshold = 0.05f;
    m->min_proportion_difference = 0.04f; m->min_prob_difference = 0.0f;
    return CHN_OK;
}

// host restatement of the model + call used to re-evaluate reads the device flags as borderline
static float host_kde_prob(const std::vector<float> &data, float h, float x) {
    float total = 0.0f;
    for (float xi : data) {
        const float t = (x - xi) / h;
        total += (float)(std::exp(-std::pow((double)t, 2) / 2) / std::sqrt(2 * 3.141592653589793238463));
    }
    return total / (h * (float)data.size());
}
static void host_model_call(const HostModel &M, uint32_t nh, const uint32_t *cnts, const uint32_t *uq, float mq, float comp,
                            uint32_t length, double *prob, uint8_t *call_out, uint8_t *conf_out) {
    const uint32_t C = M.C;
    std::vector<float> props(C), uprops(C);
    for (uint32_t c = 0; c < C; ++c) {
        props[c] = (float)cnts[c] / (float)nh;
        uprops[c] = (float)uq[c] / (float)nh;
        const float x = uprops[c];
        float p_err;
        // statslib: NaN in -> quiet_NaN out
        if (std::isnan(x)) p_err = std::numeric_limits<float>::quiet_NaN(); else if (x < 0.0f) p_err = 0.0f; else p_err = std::exp(std::log(M.rate) - M.rate * x);
        float p_pos = host_kde_prob(M.pos[c], M.h_pos, x);
        const float p_neg = host_kde_prob(M.neg[c], M.h_neg, x);
        if (x == 1.0f) p_pos = 1.0f;
        const float total = p_err + p_pos + p_neg;
        prob[c] = (double)(p_pos / total);
    }
    uint8_t call = 255; uint32_t conf;
    if (!M.paired) {
        const uint32_t host = M.host_index, other = 1u - host;
        const double hu = uprops[host], ou = uprops[other], hp = prob[host], op = prob[other];
        uint32_t first = host, second = other;
        if (hu < ou) { first = other; second = host; }
        const uint32_t raw = uq[first] - uq[second];
        conf = raw > 255u ? 255u : raw;
        if (!((int)conf < (int)M.conf_thr) && !(mq < M.min_quality) && !(length < M.min_length) && !(comp < M.min_compression)) {
            const double dc = (double)conf;
            if (hu > ou && hu - ou > M.min_pd && hp > op && hp - op > M.min_prd && std::max(hp * dc, dc) >= M.cpt) call = (uint8_t)host;
            else if (hu < M.lo_thr && hu < ou && ou - hu > M.min_pd && hp < op && op - hp > M.min_prd && std::max(op * dc, dc) >= M.cpt) call = (uint8_t)other;
        }
    } else {
        uint32_t first = 0, second = 1;
        if (uq[second] > uq[first]) std::swap(first, second);
        for (uint32_t i = 2; i < C; ++i)
            if (uq[i] > uq[second]) { second = i; if (uq[second] > uq[first]) std::swap(first, second); }
        const uint32_t raw = uq[first] - uq[second];
        conf = raw > 255u ? 255u : raw;
        if (!(mq < M.min_quality) && !(length < M.min_length) && !(comp < M.min_compression)) {
            if (prob[second] == 0 && prob[first] > 0) call = (uint8_t)first;
            else if ((int)conf > (int)M.conf_thr && prob[first] > prob[second]) call = (uint8_t)first;
            if (cnts[second] > cnts[first] || cnts[first] - cnts[second] < M.min_hits) call = 255;
            if (props[second] > props[first] || props[first] - props[second] < M.min_pd) call = 255;
        }
    }
    *call_out = call; *conf_out = (uint8_t)conf;
}

// ---- stream -------------------------------------------------------------------------------------
extern "C" int chn_stream_create(chn_index *idx, const chn_stream_cfg *cfg, chn_stream **out) {
    if (!idx || !cfg || !out || cfg->struct_size != sizeof(chn_stream_cfg)) return fail(CHN_E_INVALID, "chn_stream_create: bad argument");
    if (cfg->max_reads == 0 || cfg->max_reads > 0xFFFFFF00ULL) return fail(CHN_E_INVALID, "max_reads out of range");
    HIPCHK(hipSetDevice(idx->d.device));
    chn_stream *s = new (std::nothrow) chn_stream();
    if (!s) return fail(CHN_E_NOMEM, "host allocation failed");
    s->idx = idx; s->cfg = *cfg;
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream0, hipStreamNonBlocking);
    if (e != hipSuccess) { chn_stream_destroy(s); return fail(CHN_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    const uint64_t n = cfg->max_reads, C = idx->d.num_categories;
    int rc = CHN_OK;
    for (Slot &sl : s->slot) {
        for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipEventCreate(&sl.ev[i]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.k2_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming);
        if (e != hipSuccess) { chn_stream_destroy(s); return fail(CHN_E_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
        if ((rc = sl.d_num_hashes.ensure(n * 4)) || (rc = sl.d_counts.ensure(n * C * 4)) || (rc = sl.d_unique.ensure(n * C * 4)) ||
            (rc = sl.d_prob.ensure(n * C * 8)) || (rc = sl.d_call.ensure(n)) || (rc = sl.d_conf.ensure(n)) || (rc = sl.d_flags.ensure(n)) ||
            (rc = sl.d_acc.ensure(16))) {
            chn_stream_destroy(s);
            return rc;
        }
    }
    if ((rc = s->d_order.ensure(n * 4)) || (rc = s->d_hist.ensure(LEN_BLOCKS * 256 * 4))) { chn_stream_destroy(s); return rc; }
    const bool sharded = idx->d.row_begin != 0 || idx->d.row_end != idx->d.bin_size;
    const bool fused = !sharded && idx->single_bin_categories && C <= 8 && idx->d.bin_words == 1;
    if (!fused) {
        const uint64_t nw = (n + WAVE - 1) / WAVE;
        if ((rc = s->d_rows.ensure((cfg->max_bases + 4 * nw) * idx->d.bin_words * 8)) || (rc = s->d_rowown.ensure((cfg->max_bases + 4 * nw) * 4)) ||
            (rc = s->d_wbase.ensure(nw * 8)) || (rc = s->d_wcount.ensure(nw * 4))) { chn_stream_destroy(s); return rc; }
    }
    *out = s;
    return CHN_OK;
}

extern "C" int chn_stream_destroy(chn_stream *s) {
    if (!s) return CHN_OK;
    (void)hipSetDevice(s->idx->d.device);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); }
    if (s->stream2) { (void)hipStreamSynchronize(s->stream2); (void)hipStreamDestroy(s->stream2); }
    if (s->stream0) { (void)hipStreamSynchronize(s->stream0); (void)hipStreamDestroy(s->stream0); }
    if (s->stream) (void)hipStreamDestroy(s->stream);
    for (Slot &sl : s->slot) {
        for (int i = 0; i < 8; ++i) if (sl.ev[i]) (void)hipEventDestroy(sl.ev[i]);
        if (sl.k2_done) (void)hipEventDestroy(sl.k2_done);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        DevBuf *bufs[] = {&sl.d_num_hashes, &sl.d_counts, &sl.d_unique, &sl.d_prob, &sl.d_call, &sl.d_conf, &sl.d_flags, &sl.d_acc,
                          &sl.d_len1, &sl.d_len2, &sl.d_mq, &sl.d_comp, &sl.d_bases, &sl.d_nmask, &sl.d_off1, &sl.d_off2};
        for (DevBuf *b : bufs) b->release();
    }
    DevBuf *bufs[] = {&s->d_order, &s->d_hist, &s->d_rows, &s->d_rowown, &s->d_wbase, &s->d_wcount, &s->d_model, &s->d_list, &s->d_cbase, &s->d_memo};
    for (DevBuf *b : bufs) b->release();
    delete s;
    return CHN_OK;
}

extern "C" int chn_model_set(chn_stream *s, const chn_model *m) {
    if (!s || !m || m->struct_size != sizeof(chn_model)) return fail(CHN_E_INVALID, "chn_model_set: bad argument");
    const uint32_t C = m->num_categories;
    if (C != s->idx->d.num_categories) return fail(CHN_E_INVALID, "model category count differs from the index");
    if (!m->paired && (C < 2 || m->host_index > 1))
        return fail(CHN_E_INVALID, "single-end dehost (call_host) needs the host category at index 0 or 1 of a >= 2 category index");
    if (m->paired && C < 2) return fail(CHN_E_INVALID, "call_category needs at least 2 categories");
    HIPCHK(hipSetDevice(s->idx->d.device));
    HostModel &M = s->model;
    M = HostModel();
    M.C = C;
    size_t total = 0;
    for (uint32_t c = 0; c < C; ++c) {
        if (!m->pos_data[c] || !m->neg_data[c]) return fail(CHN_E_INVALID, "null KDE dataset");
        M.pos.emplace_back(m->pos_data[c], m->pos_data[c] + m->pos_n[c]);
        M.neg.emplace_back(m->neg_data[c], m->neg_data[c] + m->neg_n[c]);
        total += m->pos_n[c] + m->neg_n[c];
    }
    M.h_pos = m->h_pos; M.h_neg = m->h_neg; M.rate = m->err_rate;
    M.min_quality = m->min_quality; M.min_length = m->min_length; M.min_compression = m->min_compression;
    M.conf_thr = m->confidence_threshold; M.min_hits = m->min_hits; M.paired = m->paired; M.host_index = m->host_index;
    M.cpt = m->confidence_probability_threshold; M.lo_thr = m->host_unique_prop_lo_threshold;
    M.min_pd = m->min_proportion_difference; M.min_prd = m->min_prob_difference;
    std::vector<float> flat;
    flat.reserve(total);
    std::vector<uint32_t> tab(4 * (size_t)C);
    K3Args &k = s->k3;
    std::memset(&k, 0, sizeof(k));
    for (uint32_t c = 0; c < C; ++c) {
        tab[c] = (uint32_t)flat.size(); tab[C + c] = (uint32_t)M.pos[c].size();
        flat.insert(flat.end(), M.pos[c].begin(), M.pos[c].end());
        tab[2 * C + c] = (uint32_t)flat.size(); tab[3 * C + c] = (uint32_t)M.neg[c].size();
        flat.insert(flat.end(), M.neg[c].begin(), M.neg[c].end());
    }
    if (s->inflight) return fail(CHN_E_STATE, "chn_model_set: batches are in flight");
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipStreamSynchronize(s->stream2));
    const size_t tab_bytes = tab.size() * 4;
    int rc = s->d_model.ensure(tab_bytes + std::max<size_t>(flat.size() * 4, 4));
    if (rc) return rc;
    HIPCHK(hipMemcpy(s->d_model.p, tab.data(), tab_bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(static_cast<char *>(s->d_model.p) + tab_bytes, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
    const uint32_t memo_entries = 1u << 21;  // 32 MiB
    if ((rc = s->d_memo.ensure((size_t)memo_entries * 16))) return rc;
    HIPCHK(hipMemset(s->d_memo.p, 0, (size_t)memo_entries * 16));
    k.memo = s->d_memo.as<ulonglong2>(); k.memo_mask = memo_entries - 1;
    k.tab = s->d_model.as<uint32_t>();
    k.data = reinterpret_cast<const float *>(static_cast<char *>(s->d_model.p) + tab_bytes);
    k.C = C; k.h_pos = M.h_pos; k.h_neg = M.h_neg; k.rate = M.rate; k.log_rate = std::log(M.rate);
    k.min_quality = M.min_quality; k.min_compression = M.min_compression; k.cpt = M.cpt; k.lo_thr = M.lo_thr;
    k.min_pd = M.min_pd; k.min_prd = M.min_prd; k.min_length = M.min_length; k.conf_thr = (int32_t)M.conf_thr;
    k.min_hits = M.min_hits; k.paired = M.paired; k.host_index = M.host_index;
    M.set = true;
    return CHN_OK;
}

template <int W, int MODE>
static hipError_t launch_k1(const K1Args &a, size_t lds, hipStream_t st) {
    const uint32_t blocks = (a.n_reads + WAVE - 1) / WAVE;
    if (lds > 48 * 1024) {  // long windows: opt in to large dynamic LDS
        hipError_t e = hipFuncSetAttribute((const void *)k_minimise_probe<W, MODE, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (a.wn == 23)  // w=41, k=19: the defaults every real Charon index uses (include/index_arguments.hpp:16-17)
        hipLaunchKernelGGL((k_minimise_probe<W, MODE, 23>), dim3(blocks), dim3(WAVE), lds, st, a);
    else
        hipLaunchKernelGGL((k_minimise_probe<W, MODE, 0>), dim3(blocks), dim3(WAVE), lds, st, a);
    return hipGetLastError();
}
template <int MODE>
static hipError_t launch_k1_w(uint32_t W, const K1Args &a, size_t lds, hipStream_t st) {
    switch (W) {
        case 1: return launch_k1<1, MODE>(a, lds, st);
        case 2: return launch_k1<2, MODE>(a, lds, st);
        case 3: return launch_k1<3, MODE>(a, lds, st);
        default: return launch_k1<4, MODE>(a, lds, st);
    }
}
static size_t k1_lds_bytes(uint32_t wn, uint32_t C, int mode) {
    size_t b = (size_t)wn * WAVE * 8 + QCAP * 8 + QCAP * 4 + WAVE * 8 + (size_t)wn * WAVE;
    if (mode == MODE_FUSED) b += (size_t)2 * C * WAVE * 4;
    return b;
}

static int upload(DevBuf &buf, const void *src, size_t bytes, hipStream_t st) {
    int rc = buf.ensure(std::max<size_t>(bytes, 16));
    if (rc) return rc;
    if (bytes) HIPCHK(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, st));
    return CHN_OK;
}

// count (general layouts) + model/call + byte counter; closes the batch in slot `sl`
static int launch_tail(chn_stream *s, Slot &sl, bool fused);

// list_mode: stop after logging the minimiser VALUES (row-sharded mode); otherwise the whole chain
static int submit_impl(chn_stream *s, const chn_batch *b, bool list_mode) {
    if (!s || !b || b->struct_size != sizeof(chn_batch)) return fail(CHN_E_INVALID, "chn_batch_submit: bad argument");
    if (!list_mode && (s->idx->d.row_begin != 0 || s->idx->d.row_end != s->idx->d.bin_size))
        return fail(CHN_E_INVALID, "chn_batch_submit needs an index object holding all rows; use the chn_shard_* calls with a row shard");
    if (list_mode && s->inflight) return fail(CHN_E_STATE, "chn_shard_minimise: batches are in flight");
    if (s->inflight >= 2) return fail(CHN_E_STATE, "two batches already in flight: call chn_batch_wait first");
    if (b->n_reads == 0) return fail(CHN_E_INVALID, "empty batch");
    if (b->n_reads > s->cfg.max_reads || b->n_bases > s->cfg.max_bases) return fail(CHN_E_CAPACITY, "batch exceeds stream capacity");
    if (!b->bases2 || !b->seg1_offset || !b->seg1_length) return fail(CHN_E_INVALID, "missing batch arrays");
    if ((b->seg2_offset == nullptr) != (b->seg2_length == nullptr)) return fail(CHN_E_INVALID, "seg2_offset/seg2_length must both be set or both NULL");
    if (b->n_bases % 64) return fail(CHN_E_INVALID, "n_bases must be a multiple of 64");
    const chn_index_desc &d = s->idx->d;
    HIPCHK(hipSetDevice(d.device));
    const uint64_t n = b->n_reads;
    const bool paired = b->seg2_offset != nullptr;
    const uint32_t W = (uint32_t)d.bin_words, C = d.num_categories;
    const bool fused = !list_mode && s->idx->single_bin_categories && C <= 8 && W == 1;
    Slot &sl = s->slot[s->head];
    sl.host_batch = !b->on_device;
    const uint32_t *bases, *nmask;
    const uint64_t *off1, *off2;
    if (!b->on_device) {
        // host-side validation of operand shapes before anything is launched
        uint64_t total_len = 0;
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t o1 = b->seg1_offset[i], l1 = b->seg1_length[i];
            total_len += l1 + (paired ? b->seg2_length[i] : 0);
            if ((o1 & 63) || o1 + l1 > b->n_bases) return fail(CHN_E_INVALID, "segment 1 of read " + std::to_string(i) + " is misaligned or out of range");
            if (paired) {
                const uint64_t o2 = b->seg2_offset[i], l2 = b->seg2_length[i];
                if ((o2 & 63) || o2 + l2 > b->n_bases) return fail(CHN_E_INVALID, "segment 2 of read " + std::to_string(i) + " is misaligned or out of range");
            }
        }
        // the row / minimiser logs hold at most one entry per base of a segment; segments may overlap in `bases2`
        // (chunked references), so it is the SUM of the segment lengths that must fit, not n_bases
        if (total_len > s->cfg.max_bases) return fail(CHN_E_CAPACITY, "sum of segment lengths exceeds the stream's max_bases");
        int rc;
        // uploads go through the copy stream; the compute stream waits on the slot's event.  The slot's previous batch was
        // waited for before this submit could happen (at most two batches in flight), so its staging is free.
        hipStream_t cs = s->stream0;
        if ((rc = upload(sl.d_bases, b->bases2, b->n_bases / 4, cs))) return rc;
        if (b->nmask && (rc = upload(sl.d_nmask, b->nmask, b->n_bases / 8, cs))) return rc;
        if ((rc = upload(sl.d_off1, b->seg1_offset, n * 8, cs)) || (rc = upload(sl.d_len1, b->seg1_length, n * 4, cs))) return rc;
        if (paired && ((rc = upload(sl.d_off2, b->seg2_offset, n * 8, cs)) || (rc = upload(sl.d_len2, b->seg2_length, n * 4, cs)))) return rc;
        if (b->mean_quality && (rc = upload(sl.d_mq, b->mean_quality, n * 4, cs))) return rc;
        if (b->compression && (rc = upload(sl.d_comp, b->compression, n * 4, cs))) return rc;
        HIPCHK(hipEventRecord(sl.uploaded, cs));
        HIPCHK(hipStreamWaitEvent(s->stream, sl.uploaded, 0));
        bases = sl.d_bases.as<uint32_t>();
        nmask = b->nmask ? sl.d_nmask.as<uint32_t>() : nullptr;
        off1 = sl.d_off1.as<uint64_t>(); sl.len1 = sl.d_len1.as<uint32_t>();
        off2 = paired ? sl.d_off2.as<uint64_t>() : nullptr; sl.len2 = paired ? sl.d_len2.as<uint32_t>() : nullptr;
        sl.mq = b->mean_quality ? sl.d_mq.as<float>() : nullptr;
        sl.comp = b->compression ? sl.d_comp.as<float>() : nullptr;
        sl.h_len1.assign(b->seg1_length, b->seg1_length + n);
        if (paired) sl.h_len2.assign(b->seg2_length, b->seg2_length + n); else sl.h_len2.clear();
        if (b->mean_quality) sl.h_mq.assign(b->mean_quality, b->mean_quality + n); else sl.h_mq.clear();
        if (b->compression) sl.h_comp.assign(b->compression, b->compression + n); else sl.h_comp.clear();
    } else {
        bases = b->bases2; nmask = b->nmask; off1 = b->seg1_offset; sl.len1 = b->seg1_length;
        off2 = b->seg2_offset; sl.len2 = b->seg2_length; sl.mq = b->mean_quality; sl.comp = b->compression;
    }
    sl.n_reads = n;
    const bool prof = (s->cfg.flags & CHN_STREAM_PROFILE) != 0;
    for (int i = 0; i < 4; ++i) sl.ev_used[i] = false;
    if (prof) HIPCHK(hipEventRecord(sl.ev[6], s->stream));

    // 1. length-class ordering
    hipLaunchKernelGGL(k_len_hist, dim3(LEN_BLOCKS), dim3(256), 0, s->stream, sl.len1, sl.len2, (uint32_t)n, s->d_hist.as<uint32_t>());
    hipLaunchKernelGGL(k_len_scan, dim3(1), dim3(256), 0, s->stream, s->d_hist.as<uint32_t>());
    hipLaunchKernelGGL(k_len_scatter, dim3(LEN_BLOCKS), dim3(256), 0, s->stream, sl.len1, sl.len2, (uint32_t)n, s->d_hist.as<uint32_t>(), s->d_order.as<uint32_t>());
    HIPCHK(hipGetLastError());

    // 2. minimise + probe
    K1Args a;
    std::memset(&a, 0, sizeof(a));
    a.words = s->idx->words; a.S = d.bin_size; a.row_begin = d.row_begin; a.row_end = d.row_end; a.seed = d.minimiser_seed; a.powk1 = pow5(d.kmer_size - 1);
    a.shift = (uint32_t)d.hash_shift; a.h = d.hash_funs; a.k = d.kmer_size; a.wn = d.window_size - d.kmer_size + 1;
    a.n_reads = (uint32_t)n; a.nseg = paired ? 2 : 1; a.B = (uint32_t)d.bins; a.C = C;
    for (uint32_t bb = 0; bb < 8 && bb < d.bins; ++bb) a.b2c_packed |= (uint64_t)d.bin_to_category[bb] << (8 * bb);
    a.bases = bases; a.nmask = nmask; a.off1 = off1; a.off2 = off2; a.len1 = sl.len1; a.len2 = sl.len2;
    a.order = s->d_order.as<uint32_t>();
    if (const char *ab = std::getenv("CHN_ABLATE")) a.ablate = (uint32_t)std::atoi(ab);
    a.nt_probes = s->idx->rows_local * d.bin_words * 8 > (4ULL << 30) ? 1u : 0u;
    if (const char *nt = std::getenv("CHN_NT_PROBES")) a.nt_probes = (uint32_t)std::atoi(nt);  // A/B diagnostic
    a.num_hashes = sl.d_num_hashes.as<uint32_t>(); a.counts = sl.d_counts.as<uint32_t>(); a.unique = sl.d_unique.as<uint32_t>();
    a.rows = list_mode ? s->d_list.as<uint64_t>() : s->d_rows.as<uint64_t>(); a.rowlog = s->d_rowown.as<uint32_t>();
    a.wave_base = s->d_wbase.as<uint64_t>(); a.wave_count = s->d_wcount.as<uint32_t>();
    const uint32_t n_waves = (uint32_t)((n + WAVE - 1) / WAVE);
    if (!fused) {
        hipLaunchKernelGGL(k_wave_caps, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s->stream, s->d_order.as<uint32_t>(), sl.len1, sl.len2,
                           (uint32_t)n, s->d_wbase.as<uint64_t>());
        hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, s->stream, s->d_wbase.as<uint64_t>(), n_waves);
        HIPCHK(hipGetLastError());
    }
    const int mode = list_mode ? MODE_LIST : fused ? MODE_FUSED : MODE_ROWS;
    size_t lds = k1_lds_bytes(a.wn, C, mode);
    if (const char *pad = std::getenv("CHN_LDS_PAD")) lds += (size_t)std::atoi(pad);  // occupancy-sensitivity diagnostic
    if (lds > 160 * 1024) return fail(CHN_E_INVALID, "window too large for LDS");
    if (prof) { HIPCHK(hipEventRecord(sl.ev[0], s->stream)); }
    hipError_t e = list_mode ? launch_k1<1, MODE_LIST>(a, lds, s->stream)
                             : fused ? launch_k1_w<MODE_FUSED>(W, a, lds, s->stream) : launch_k1_w<MODE_ROWS>(W, a, lds, s->stream);
    if (e != hipSuccess) return fail(CHN_E_HIP, std::string("k_minimise_probe launch: ") + hipGetErrorString(e));
    if (prof) { HIPCHK(hipEventRecord(sl.ev[1], s->stream)); sl.ev_used[0] = true; }

    if (list_mode) return CHN_OK;  // the caller continues with chn_shard_probe / chn_shard_finish
    return launch_tail(s, sl, fused);
}

static int launch_tail(chn_stream *s, Slot &sl, bool fused) {
    const chn_index_desc &d = s->idx->d;
    const uint64_t n = sl.n_reads;
    const uint32_t W = (uint32_t)d.bin_words, C = d.num_categories;
    const uint32_t n_waves = (uint32_t)((n + WAVE - 1) / WAVE);
    const bool prof = (s->cfg.flags & CHN_STREAM_PROFILE) != 0;
    // 3. counts from rows (general bin layouts)
    if (!fused) {
        K2Args k2;
        std::memset(&k2, 0, sizeof(k2));
        k2.rows = s->d_rows.as<uint64_t>(); k2.rowlog = s->d_rowown.as<uint32_t>();
        k2.wave_base = s->d_wbase.as<uint64_t>(); k2.wave_count = s->d_wcount.as<uint32_t>(); k2.order = s->d_order.as<uint32_t>();
        k2.counts = sl.d_counts.as<uint32_t>(); k2.unique = sl.d_unique.as<uint32_t>();
        k2.n_reads = (uint32_t)n; k2.B = (uint32_t)d.bins; k2.C = C; k2.W = W;
        std::memcpy(k2.b2c, d.bin_to_category, 256);
        const dim3 grid(n_waves), block(K2_THREADS);
        const size_t lds2 = k2_lds_bytes((uint32_t)d.bins, C, W);
        if (lds2 > 150 * 1024) return fail(CHN_E_INVALID, "bins x categories too large for the count kernel's LDS");
        if (lds2 > 48 * 1024) {  // opt in to large dynamic LDS
            const void *fn = W == 1 ? (const void *)k_count_wavelog<1> : W == 2 ? (const void *)k_count_wavelog<2> : W == 3 ? (const void *)k_count_wavelog<3> : (const void *)k_count_wavelog<4>;
            HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        }
        if (prof) HIPCHK(hipEventRecord(sl.ev[2], s->stream));
        switch (W) {
            case 1: hipLaunchKernelGGL(k_count_wavelog<1>, grid, block, lds2, s->stream, k2); break;
            case 2: hipLaunchKernelGGL(k_count_wavelog<2>, grid, block, lds2, s->stream, k2); break;
            case 3: hipLaunchKernelGGL(k_count_wavelog<3>, grid, block, lds2, s->stream, k2); break;
            default: hipLaunchKernelGGL(k_count_wavelog<4>, grid, block, lds2, s->stream, k2); break;
        }
        HIPCHK(hipGetLastError());
        if (prof) { HIPCHK(hipEventRecord(sl.ev[3], s->stream)); sl.ev_used[1] = true; }
    }
    HIPCHK(hipEventRecord(sl.k2_done, s->stream));

    // 4. model + call on the side stream (overlaps the next batch's minimise+probe)
    HIPCHK(hipStreamWaitEvent(s->stream2, sl.k2_done, 0));
    sl.model_ran = s->model.set;
    if (s->model.set) {
        K3Args k3 = s->k3;
        k3.num_hashes = sl.d_num_hashes.as<uint32_t>(); k3.counts = sl.d_counts.as<uint32_t>(); k3.unique = sl.d_unique.as<uint32_t>();
        k3.len1 = sl.len1; k3.len2 = sl.len2; k3.mean_quality = sl.mq; k3.compression = sl.comp;
        k3.prob = sl.d_prob.as<double>(); k3.call = sl.d_call.as<uint8_t>(); k3.conf = sl.d_conf.as<uint8_t>(); k3.flags = sl.d_flags.as<uint8_t>();
        k3.n_reads = (uint32_t)n;
        if (prof) HIPCHK(hipEventRecord(sl.ev[4], s->stream2));
        hipLaunchKernelGGL(k_model_call, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s->stream2, k3);
        HIPCHK(hipGetLastError());
        if (prof) { HIPCHK(hipEventRecord(sl.ev[5], s->stream2)); sl.ev_used[2] = true; }
    }
    if (prof) { HIPCHK(hipEventRecord(sl.ev[7], s->stream2)); sl.ev_used[3] = true; }
    // algorithmic byte count of this batch (outside the profiled chain)
    HIPCHK(hipMemsetAsync(sl.d_acc.p, 0, 16, s->stream2));
    hipLaunchKernelGGL(k_batch_bytes, dim3(std::min<uint64_t>(1024, (n + 255) / 256)), dim3(256), 0, s->stream2, sl.len1, sl.len2,
                       sl.d_num_hashes.as<uint32_t>(), (uint32_t)n, (uint32_t)(d.hash_funs * W * 8), (uint32_t)(8 + 8 * C),
                       sl.d_acc.as<unsigned long long>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(sl.done, s->stream2));
    s->head ^= 1;
    s->inflight += 1;
    return CHN_OK;
}

extern "C" int chn_batch_submit(chn_stream *s, const chn_batch *b) { return submit_impl(s, b, false); }

// ---- row-sharded mode -----------------------------------------------------------------------------
static int ensure_shard_buffers(chn_stream *s) {
    const uint64_t nw = (s->cfg.max_reads + WAVE - 1) / WAVE;
    int rc;
    if ((rc = s->d_list.ensure((s->cfg.max_bases + 4 * nw) * 8)) || (rc = s->d_cbase.ensure((nw + 1) * 8))) return rc;
    if ((rc = s->d_rows.ensure((s->cfg.max_bases + 4 * nw) * s->idx->d.bin_words * 8)) || (rc = s->d_rowown.ensure((s->cfg.max_bases + 4 * nw) * 4)) ||
        (rc = s->d_wbase.ensure(nw * 8)) || (rc = s->d_wcount.ensure(nw * 4))) return rc;
    return CHN_OK;
}
static ShardArgs shard_args(chn_stream *s, const chn_index *shard, uint64_t *partial) {
    ShardArgs a;
    std::memset(&a, 0, sizeof a);
    const chn_index_desc &d = shard->d;
    a.list = s->d_list.as<uint64_t>(); a.wave_base = s->d_wbase.as<uint64_t>(); a.cbase = s->d_cbase.as<uint64_t>();
    a.wave_count = s->d_wcount.as<uint32_t>(); a.words = shard->words; a.S = d.bin_size; a.row_begin = d.row_begin; a.row_end = d.row_end;
    a.shift = (uint32_t)d.hash_shift; a.h = d.hash_funs; a.W = (uint32_t)d.bin_words; a.partial = partial; a.rows = s->d_rows.as<uint64_t>();
    a.rowlog = s->d_rowown.as<uint32_t>(); a.B = (uint32_t)d.bins;
    return a;
}

extern "C" int chn_shard_minimise(chn_stream *s, const chn_batch *b, uint64_t *n_entries) {
    if (!s || !b || !n_entries) return fail(CHN_E_INVALID, "chn_shard_minimise: null argument");
    if (s->shard_open) return fail(CHN_E_STATE, "chn_shard_minimise: previous sharded batch not finished");
    HIPCHK(hipSetDevice(s->idx->d.device));
    int rc = ensure_shard_buffers(s);
    if (rc) return rc;
    if ((rc = submit_impl(s, b, true))) return rc;
    const uint32_t n_waves = (uint32_t)((b->n_reads + WAVE - 1) / WAVE);
    hipLaunchKernelGGL(k_counts_to_u64, dim3((n_waves + 1 + 255) / 256), dim3(256), 0, s->stream, s->d_wcount.as<uint32_t>(), n_waves, s->d_cbase.as<uint64_t>());
    hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, s->stream, s->d_cbase.as<uint64_t>(), n_waves + 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    uint64_t total = 0;
    HIPCHK(hipMemcpy(&total, s->d_cbase.as<uint64_t>() + n_waves, 8, hipMemcpyDeviceToHost));
    s->shard_entries = total;
    s->shard_open = true;
    *n_entries = total;
    return CHN_OK;
}

extern "C" int chn_minimisers(chn_stream *s, const chn_batch *b, uint64_t *host_values, uint64_t capacity, uint64_t *n_values) {
    if (!s || !b || !n_values) return fail(CHN_E_INVALID, "chn_minimisers: null argument");
    uint64_t total = 0;
    int rc = chn_shard_minimise(s, b, &total);
    if (rc) return rc;
    s->shard_open = false;  // nothing else of the sharded chain follows
    *n_values = total;
    if (total == 0) return CHN_OK;
    if (!host_values || capacity < total) return fail(CHN_E_CAPACITY, "chn_minimisers: output buffer too small (need " + std::to_string(total) + " values)");
    // compact the per-wavefront logs into the (unused here) row buffer, then copy out
    if ((rc = s->d_rows.ensure(total * 8))) return rc;
    const Slot &sl = s->slot[s->head];
    const uint32_t n_waves = (uint32_t)((sl.n_reads + WAVE - 1) / WAVE);
    const ShardArgs sa = shard_args(s, s->idx, nullptr);
    hipLaunchKernelGGL(k_compact_list, dim3(n_waves), dim3(256), 0, s->stream, sa, s->d_rows.as<uint64_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_values, s->d_rows.p, total * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CHN_OK;
}

extern "C" int chn_index_emplace(chn_index *idx, const uint64_t *host_values, uint64_t n_values, uint32_t bin) {
    if (!idx || (!host_values && n_values)) return fail(CHN_E_INVALID, "chn_index_emplace: null argument");
    if (bin >= idx->d.bins) return fail(CHN_E_INVALID, "chn_index_emplace: bin out of range");
    if (n_values == 0) return CHN_OK;
    const chn_index_desc &d = idx->d;
    HIPCHK(hipSetDevice(d.device));
    uint64_t *dv = nullptr;
    const uint64_t chunk = 1ULL << 26;  // 512 MiB of values at a time
    HIPCHK(hipMalloc((void **)&dv, std::min(chunk, n_values) * 8));
    for (uint64_t o = 0; o < n_values; o += chunk) {
        const uint64_t m = std::min(chunk, n_values - o);
        hipError_t e = hipMemcpy(dv, host_values + o, m * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_emplace_values, dim3((uint32_t)std::min<uint64_t>(65535, (m + 255) / 256)), dim3(256), 0, 0, idx->words, dv, m, bin,
                               d.bin_size, d.row_begin, d.row_end, (uint32_t)d.hash_shift, (uint32_t)d.hash_funs, (uint32_t)d.bin_words);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { (void)hipFree(dv); return fail(CHN_E_HIP, std::string("chn_index_emplace: ") + hipGetErrorString(e)); }
    }
    HIPCHK(hipFree(dv));
    return CHN_OK;
}

extern "C" int chn_shard_probe(chn_stream *s, const chn_index *shard, uint64_t *dev_partial, uint64_t capacity_words) {
    if (!s || !shard || !dev_partial) return fail(CHN_E_INVALID, "chn_shard_probe: null argument");
    if (!s->shard_open) return fail(CHN_E_STATE, "chn_shard_probe: call chn_shard_minimise first");
    const chn_index_desc &a = s->idx->d, &c = shard->d;
    if (a.bin_size != c.bin_size || a.bin_words != c.bin_words || a.hash_funs != c.hash_funs || a.device != c.device)
        return fail(CHN_E_INVALID, "chn_shard_probe: shard does not belong to the stream's index");
    if (capacity_words < s->shard_entries * c.hash_funs * c.bin_words) return fail(CHN_E_CAPACITY, "chn_shard_probe: partial buffer too small");
    HIPCHK(hipSetDevice(a.device));
    const Slot &sl = s->slot[s->head];
    const uint32_t n_waves = (uint32_t)((sl.n_reads + WAVE - 1) / WAVE);
    const ShardArgs sa = shard_args(s, shard, dev_partial);
    switch (c.bin_words) {
        case 1: hipLaunchKernelGGL(k_probe_partial<1>, dim3(n_waves), dim3(256), 0, s->stream, sa); break;
        case 2: hipLaunchKernelGGL(k_probe_partial<2>, dim3(n_waves), dim3(256), 0, s->stream, sa); break;
        case 3: hipLaunchKernelGGL(k_probe_partial<3>, dim3(n_waves), dim3(256), 0, s->stream, sa); break;
        default: hipLaunchKernelGGL(k_probe_partial<4>, dim3(n_waves), dim3(256), 0, s->stream, sa); break;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));  // the caller's collective runs on its own stream
    return CHN_OK;
}

extern "C" int chn_shard_finish(chn_stream *s, const uint64_t *dev_partial) {
    if (!s || !dev_partial) return fail(CHN_E_INVALID, "chn_shard_finish: null argument");
    if (!s->shard_open) return fail(CHN_E_STATE, "chn_shard_finish: no sharded batch open");
    HIPCHK(hipSetDevice(s->idx->d.device));
    Slot &sl = s->slot[s->head];
    const uint32_t n_waves = (uint32_t)((sl.n_reads + WAVE - 1) / WAVE);
    const ShardArgs sa = shard_args(s, s->idx, const_cast<uint64_t *>(dev_partial));
    switch (s->idx->d.bin_words) {
        case 1: hipLaunchKernelGGL(k_and_partial<1>, dim3(n_waves), dim3(256), 0, s->stream, sa); break;
        case 2: hipLaunchKernelGGL(k_and_partial<2>, dim3(n_waves), dim3(256), 0, s->stream, sa); break;
        case 3: hipLaunchKernelGGL(k_and_partial<3>, dim3(n_waves), dim3(256), 0, s->stream, sa); break;
        default: hipLaunchKernelGGL(k_and_partial<4>, dim3(n_waves), dim3(256), 0, s->stream, sa); break;
    }
    HIPCHK(hipGetLastError());
    s->shard_open = false;
    return launch_tail(s, sl, false);
}

extern "C" int chn_stream_sync(chn_stream *s) {
    if (!s) return fail(CHN_E_INVALID, "null stream");
    HIPCHK(hipSetDevice(s->idx->d.device));
    HIPCHK(hipStreamSynchronize(s->stream0));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipStreamSynchronize(s->stream2));
    return CHN_OK;
}

extern "C" int chn_batch_wait(chn_stream *s, chn_result *r) {
    if (!s || !r || r->struct_size != sizeof(chn_result)) return fail(CHN_E_INVALID, "chn_batch_wait: bad argument");
    if (s->inflight == 0) return fail(CHN_E_STATE, "no batch in flight");
    HIPCHK(hipSetDevice(s->idx->d.device));
    Slot &sl = s->slot[s->inflight == 2 ? s->head : (s->head ^ 1)];  // oldest batch in flight
    HIPCHK(hipEventSynchronize(sl.done));
    s->inflight -= 1;
    static const int evpair[4][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}};
    for (int i = 0; i < 4; ++i)
        if (sl.ev_used[i]) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, sl.ev[evpair[i][0]], sl.ev[evpair[i][1]]));
            s->prof_ms[i] += ms; s->prof_n[i] += 1;
        }
    unsigned long long acc[2];
    HIPCHK(hipMemcpy(acc, sl.d_acc.p, 16, hipMemcpyDeviceToHost));
    s->last_bytes = acc[0]; s->last_min = acc[1];
    const uint64_t n = sl.n_reads, C = s->idx->d.num_categories;
    if (r->on_device) {  // valid until the second-next chn_batch_submit
        r->num_hashes = sl.d_num_hashes.as<uint32_t>(); r->counts = sl.d_counts.as<uint32_t>(); r->unique_counts = sl.d_unique.as<uint32_t>();
        r->probabilities = sl.d_prob.as<double>(); r->call = sl.d_call.as<uint8_t>(); r->confidence = sl.d_conf.as<uint8_t>();
        r->flags = sl.d_flags.as<uint8_t>();
        return CHN_OK;
    }
    if (r->num_hashes) HIPCHK(hipMemcpy(r->num_hashes, sl.d_num_hashes.p, n * 4, hipMemcpyDeviceToHost));
    if (r->counts) HIPCHK(hipMemcpy(r->counts, sl.d_counts.p, n * C * 4, hipMemcpyDeviceToHost));
    if (r->unique_counts) HIPCHK(hipMemcpy(r->unique_counts, sl.d_unique.p, n * C * 4, hipMemcpyDeviceToHost));
    if (sl.model_ran) {
        if (r->probabilities) HIPCHK(hipMemcpy(r->probabilities, sl.d_prob.p, n * C * 8, hipMemcpyDeviceToHost));
        if (r->call) HIPCHK(hipMemcpy(r->call, sl.d_call.p, n, hipMemcpyDeviceToHost));
        if (r->confidence) HIPCHK(hipMemcpy(r->confidence, sl.d_conf.p, n, hipMemcpyDeviceToHost));
        std::vector<uint8_t> flags(n);
        HIPCHK(hipMemcpy(flags.data(), sl.d_flags.p, n, hipMemcpyDeviceToHost));
        if (r->flags) std::memcpy(r->flags, flags.data(), n);
        // borderline reads: re-evaluate with the host libm so that `call` never depends on a last-ulp exp() difference
        if (sl.host_batch && r->num_hashes && r->counts && r->unique_counts && r->call && r->confidence && r->probabilities) {
            std::vector<double> p(C);
            for (uint64_t i = 0; i < n; ++i)
                if (flags[i] || r->num_hashes[i] == 0) {
                    const uint32_t length = sl.h_len1[i] + (sl.h_len2.empty() ? 0u : sl.h_len2[i]);
                    host_model_call(s->model, r->num_hashes[i], r->counts + i * C, r->unique_counts + i * C,
                                    sl.h_mq.empty() ? 0.0f : sl.h_mq[i], sl.h_comp.empty() ? 0.0f : sl.h_comp[i], length, p.data(),
                                    &r->call[i], &r->confidence[i]);
                    for (uint64_t c = 0; c < C; ++c) r->probabilities[i * C + c] = p[c];
                }
        }
    }
    return CHN_OK;
}

extern "C" int chn_classify_counts(chn_stream *s, uint64_t n, const uint32_t *num_hashes, const uint32_t *counts,
                                   const uint32_t *unique_counts, const uint32_t *lengths, const float *mean_quality,
                                   const float *compression, double *probabilities, uint8_t *call, uint8_t *confidence) {
    if (!s || !num_hashes || !counts || !unique_counts || !lengths || !probabilities || !call || !confidence)
        return fail(CHN_E_INVALID, "chn_classify_counts: null argument");
    if (!s->model.set) return fail(CHN_E_STATE, "chn_classify_counts: no model set");
    if (s->inflight) return fail(CHN_E_STATE, "chn_classify_counts: a batch is in flight");
    Slot &sl = s->slot[0];
    if (n == 0) return CHN_OK;
    if (n > s->cfg.max_reads) return fail(CHN_E_CAPACITY, "chn_classify_counts: more reads than the stream's max_reads");
    HIPCHK(hipSetDevice(s->idx->d.device));
    const uint64_t C = s->idx->d.num_categories;
    int rc;
    if ((rc = upload(sl.d_len1, lengths, n * 4, s->stream))) return rc;
    if (mean_quality && (rc = upload(sl.d_mq, mean_quality, n * 4, s->stream))) return rc;
    if (compression && (rc = upload(sl.d_comp, compression, n * 4, s->stream))) return rc;
    HIPCHK(hipMemcpyAsync(sl.d_num_hashes.p, num_hashes, n * 4, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(sl.d_counts.p, counts, n * C * 4, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(sl.d_unique.p, unique_counts, n * C * 4, hipMemcpyHostToDevice, s->stream));
    K3Args k3 = s->k3;
    k3.num_hashes = sl.d_num_hashes.as<uint32_t>(); k3.counts = sl.d_counts.as<uint32_t>(); k3.unique = sl.d_unique.as<uint32_t>();
    k3.len1 = sl.d_len1.as<uint32_t>(); k3.len2 = nullptr;
    k3.mean_quality = mean_quality ? sl.d_mq.as<float>() : nullptr; k3.compression = compression ? sl.d_comp.as<float>() : nullptr;
    k3.prob = sl.d_prob.as<double>(); k3.call = sl.d_call.as<uint8_t>(); k3.conf = sl.d_conf.as<uint8_t>(); k3.flags = sl.d_flags.as<uint8_t>();
    k3.n_reads = (uint32_t)n;
    hipLaunchKernelGGL(k_model_call, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s->stream, k3);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(probabilities, sl.d_prob.p, n * C * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(call, sl.d_call.p, n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(confidence, sl.d_conf.p, n, hipMemcpyDeviceToHost));
    std::vector<uint8_t> flags(n);
    HIPCHK(hipMemcpy(flags.data(), sl.d_flags.p, n, hipMemcpyDeviceToHost));
    std::vector<double> p(C);
    for (uint64_t i = 0; i < n; ++i)
        if (flags[i] || num_hashes[i] == 0) {  // borderline or NaN rows: host libm / host NaN sign
            host_model_call(s->model, num_hashes[i], counts + i * C, unique_counts + i * C, mean_quality ? mean_quality[i] : 0.0f,
                            compression ? compression[i] : 0.0f, lengths[i], p.data(), &call[i], &confidence[i]);
            for (uint64_t c = 0; c < C; ++c) probabilities[i * C + c] = p[c];
        }
    return CHN_OK;
}

extern "C" int chn_stream_profile(chn_stream *s, int which, double *total_ms, uint64_t *launches, int reset) {
    if (!s || which < 0 || which > 3) return fail(CHN_E_INVALID, "chn_stream_profile: bad argument");
    if (total_ms) *total_ms = s->prof_ms[which];
    if (launches) *launches = s->prof_n[which];
    if (reset) { s->prof_ms[which] = 0; s->prof_n[which] = 0; }
    return CHN_OK;
}
extern "C" int chn_stream_last_batch_bytes(chn_stream *s, uint64_t *bytes, uint64_t *total_minimisers) {
    if (!s) return fail(CHN_E_INVALID, "null stream");
    if (bytes) *bytes = s->last_bytes;
    if (total_minimisers) *total_minimisers = s->last_min;
    return CHN_OK;
}

// ---- synthetic workloads ------------------------------------------------------------------------
extern "C" int chn_synth_genomes(int device, uint64_t seed, uint64_t n_genomes, uint64_t genome_len, uint32_t **dev_bases2) {
    if (!dev_bases2 || genome_len % 64 || n_genomes == 0) return fail(CHN_E_INVALID, "chn_synth_genomes: bad argument");
    HIPCHK(hipSetDevice(device));
    const uint64_t nd = n_genomes * genome_len / 16;
    uint32_t *p = nullptr;
    HIPCHK(hipMalloc((void **)&p, nd * 4));
    hipLaunchKernelGGL(k_synth_genomes, dim3((uint32_t)std::min<uint64_t>(65535, (nd + 255) / 256)), dim3(256), 0, 0, p, nd, seed);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    *dev_bases2 = p;
    return CHN_OK;
}
extern "C" int chn_synth_fill_index(chn_index *idx, uint64_t seed, double density) {
    if (!idx || density < 0 || density > 1) return fail(CHN_E_INVALID, "chn_synth_fill_index: bad argument");
    HIPCHK(hipSetDevice(idx->d.device));
    const uint32_t thr = (uint32_t)std::lround(density * 65536.0);
    hipLaunchKernelGGL(k_synth_fill, dim3(8192), dim3(256), 0, 0, idx->words, idx->rows_local, idx->d.row_begin, (uint32_t)idx->d.bin_words, (uint32_t)idx->d.bins, seed, thr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return CHN_OK;
}
extern "C" int chn_synth_plant(chn_index *idx, const uint32_t *dev_bases2, uint64_t n_genomes, uint64_t genome_len, const uint8_t *genome_bin) {
    if (!idx || !dev_bases2 || !genome_bin || genome_len % 64) return fail(CHN_E_INVALID, "chn_synth_plant: bad argument");
    const chn_index_desc &d = idx->d;
    HIPCHK(hipSetDevice(d.device));
    // The set of minimisers of a sequence equals the union over overlapping chunks (every window lies in one chunk),
    // so genomes are cut into chunks of 4096 bases overlapping by w-1 and each chunk is treated as one "read".
    const uint32_t chunk = 4096, overlap = d.window_size - 1;
    const uint64_t cpg = (genome_len + chunk - 1) / chunk, n = n_genomes * cpg;
    if (n > 0xFFFFFF00ULL) return fail(CHN_E_INVALID, "too many chunks");
    uint64_t *off = nullptr; uint32_t *len = nullptr; uint8_t *bin = nullptr;
    HIPCHK(hipMalloc((void **)&off, n * 8));
    HIPCHK(hipMalloc((void **)&len, n * 4));
    HIPCHK(hipMalloc((void **)&bin, n));
    std::vector<uint8_t> hb(n);
    for (uint64_t i = 0; i < n; ++i) {
        hb[i] = genome_bin[i / cpg];
        if (hb[i] >= d.bins) { (void)hipFree(off); (void)hipFree(len); (void)hipFree(bin); return fail(CHN_E_INVALID, "genome_bin out of range"); }
    }
    HIPCHK(hipMemcpy(bin, hb.data(), n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_chunk_layout, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, n_genomes, genome_len, chunk, overlap, cpg, off, len);
    K1Args a;
    std::memset(&a, 0, sizeof(a));
    a.words = idx->words; a.words_rw = idx->words; a.S = d.bin_size; a.row_begin = d.row_begin; a.row_end = d.row_end; a.seed = d.minimiser_seed; a.powk1 = pow5(d.kmer_size - 1);
    a.shift = (uint32_t)d.hash_shift; a.h = d.hash_funs; a.k = d.kmer_size; a.wn = d.window_size - d.kmer_size + 1;
    a.n_reads = (uint32_t)n; a.nseg = 1; a.B = (uint32_t)d.bins; a.C = d.num_categories;
    a.bases = dev_bases2; a.off1 = off; a.len1 = len; a.read_bin = bin;
    const size_t lds = k1_lds_bytes(a.wn, d.num_categories, MODE_EMPLACE);
    hipError_t e = launch_k1_w<MODE_EMPLACE>((uint32_t)d.bin_words, a, lds, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(off); (void)hipFree(len); (void)hipFree(bin);
    if (e != hipSuccess) return fail(CHN_E_HIP, std::string("plant: ") + hipGetErrorString(e));
    return CHN_OK;
}
extern "C" int chn_synth_reads(int device, uint64_t seed, const uint32_t *dev_genomes, uint64_t n_genomes, uint64_t genome_len,
                               uint64_t first_read_id, uint64_t n_reads, uint32_t read_len_min, uint32_t read_len_max, double sub_rate, double random_fraction,
                               float mean_quality, chn_synth_reads_out *out) {
    if (!out || n_reads == 0 || read_len_min == 0 || read_len_max < read_len_min) return fail(CHN_E_INVALID, "chn_synth_reads: bad argument");
    if (n_genomes && (!dev_genomes || genome_len < read_len_max)) return fail(CHN_E_INVALID, "genomes shorter than reads");
    HIPCHK(hipSetDevice(device));
    std::memset(out, 0, sizeof(*out));
    uint64_t *d_total = nullptr;
    HIPCHK(hipMalloc((void **)&out->seg1_length, n_reads * 4));
    HIPCHK(hipMalloc((void **)&out->seg1_offset, n_reads * 8));
    HIPCHK(hipMalloc((void **)&out->mean_quality, n_reads * 4));
    HIPCHK(hipMalloc((void **)&out->compression, n_reads * 4));
    HIPCHK(hipMalloc((void **)&d_total, 8));
    const dim3 g1((uint32_t)((n_reads + 255) / 256)), b1(256);
    hipLaunchKernelGGL(k_synth_read_layout, g1, b1, 0, 0, seed, first_read_id, n_reads, read_len_min, read_len_max, out->seg1_length);
    uint64_t total = 0;
    if (read_len_min == read_len_max) {
        const uint64_t pad = ((uint64_t)read_len_min + 63) & ~63ULL;
        std::vector<uint64_t> h(n_reads);
        for (uint64_t i = 0; i < n_reads; ++i) h[i] = i * pad;
        HIPCHK(hipMemcpy(out->seg1_offset, h.data(), n_reads * 8, hipMemcpyHostToDevice));
        total = n_reads * pad;
    } else {
        hipLaunchKernelGGL(k_synth_offsets, dim3(1), dim3(64), 0, 0, out->seg1_length, n_reads, out->seg1_offset, d_total);
        HIPCHK(hipMemcpy(&total, d_total, 8, hipMemcpyDeviceToHost));
    }
    (void)hipFree(d_total);
    out->n_bases = total;
    HIPCHK(hipMalloc((void **)&out->bases2, total / 4 + 16));
    HIPCHK(hipMemset(out->bases2, 0, total / 4 + 16));
    hipLaunchKernelGGL(k_fill_f32, g1, b1, 0, 0, out->mean_quality, n_reads, mean_quality);
    hipLaunchKernelGGL(k_fill_f32, g1, b1, 0, 0, out->compression, n_reads, 0.3f);
    SynthReadsArgs a;
    a.genomes = dev_genomes; a.n_genomes = n_genomes; a.genome_len = genome_len; a.seed = seed; a.n_reads = n_reads; a.first_id = first_read_id;
    a.len_min = read_len_min; a.len_max = read_len_max;
    a.sub_thr32 = (uint32_t)std::min<double>(4294967295.0, sub_rate * 4294967296.0);
    a.rand_thr32 = (uint32_t)std::min<double>(4294967295.0, random_fraction * 4294967296.0);
    a.bases = out->bases2; a.off = out->seg1_offset; a.len = out->seg1_length;
    const uint32_t gy = (uint32_t)std::min<uint64_t>(n_reads, 32768), gz = (uint32_t)((n_reads + gy - 1) / gy);
    const uint32_t max_dwords = (read_len_max + 63) / 64 * 4;
    hipLaunchKernelGGL(k_synth_reads, dim3((max_dwords + 63) / 64, gy, gz), dim3(64), 0, 0, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return CHN_OK;
}
extern "C" int chn_host_alloc(uint64_t bytes, void **ptr) {
    if (!ptr) return fail(CHN_E_INVALID, "null argument");
    HIPCHK(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return CHN_OK;
}
extern "C" int chn_host_free(void *ptr) {
    if (ptr) HIPCHK(hipHostFree(ptr));
    return CHN_OK;
}
extern "C" int chn_device_malloc(int device, uint64_t bytes, void **ptr) {
    if (!ptr) return fail(CHN_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMalloc(ptr, bytes ? bytes : 16));
    return CHN_OK;
}
extern "C" int chn_device_upload(int device, void *dev_dst, const void *host_src, uint64_t bytes) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return CHN_OK;
}
extern "C" int chn_device_free(int device, void *ptr) {
    HIPCHK(hipSetDevice(device));
    if (ptr) HIPCHK(hipFree(ptr));
    return CHN_OK;
}
extern "C" int chn_device_download(int device, void *host_dst, const void *dev_src, uint64_t bytes) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return CHN_OK;
}
