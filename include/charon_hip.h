/* charon_hip.h -- C ABI of libcharon_hip.so: the MI355X (gfx950) implementation of the per-read
 * classification path of `charon dehost`.
 *
 * The reference (rmcolq/charon) has no FFI/plugin interface; the seam this library replaces is the body
 * of the OpenMP loop in src/dehost_main.cpp:366-373 (single-end) / :456-468 (paired) plus the model
 * application and call that Result::classify_read performs on each entry (include/result.hpp:97-116 ->
 * include/read_entry.hpp:218-291).  Each entry point below cites the reference code it stands in for.
 * All paths are relative to the reference checkout.
 *
 * Conventions: every function returns 0 on success or a negative CHN_E_* code; chn_last_error() returns a
 * thread-local message owned by the library.  No C++ types, exceptions or torch types cross this boundary.
 * All sizes are 64-bit.  The library never writes to stdout/stderr.  A chn_stream is NOT thread-safe;
 * different streams may be driven from different host threads.
 */
#ifndef CHARON_HIP_H
#define CHARON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHN_OK 0
#define CHN_E_INVALID (-1)   /* bad argument / unsupported parameter combination */
#define CHN_E_HIP (-2)       /* a HIP runtime call failed (message has the HIP error string) */
#define CHN_E_NOMEM (-3)     /* device or host allocation failed */
#define CHN_E_STATE (-4)     /* call sequence error (e.g. wait without submit) */
#define CHN_E_CAPACITY (-5)  /* batch exceeds the stream's configured capacity */

#define CHN_MAX_CATEGORIES 255
#define CHN_NO_CALL 255u

typedef struct chn_index chn_index;   /* immutable IBF resident in HBM */
typedef struct chn_stream chn_stream; /* one HIP stream + its device scratch */

/* ---- index ------------------------------------------------------------------------------------------
 * Replaces: Index (include/index.hpp:19-138) as far as the hot path reads it -- window_size()/kmer_size()
 * (:52-58), the IBF parameters and data (:26, seqan3::interleaved_bloom_filter<compressed>), the bin ->
 * category map (include/input_summary.hpp:20,39-45) and get_host_index() (include/index.hpp:72-80).
 * Device layout: plain interleaved 64-bit words, word (row * bin_words + b) holds technical bins
 * 64b..64b+63 of row `row` -- the same addressing seqan3 uses (bit index row * technical_bins + bin), with
 * the Elias-Fano (sdsl::sd_vector) compression undone once at load time. */
typedef struct chn_index_desc {
    uint32_t struct_size;      /* sizeof(chn_index_desc) */
    int32_t device;            /* HIP device ordinal */
    uint8_t kmer_size;         /* k  (1..27: 5^k must fit 64 bits, as in seqan3 for dna5) */
    uint8_t window_size;       /* w >= k */
    uint8_t hash_funs;         /* h in 1..5 */
    uint8_t num_categories;    /* C */
    uint8_t host_index;        /* InputSummary::host_category_index(); 255 if none */
    uint8_t reserved0[3];
    uint64_t minimiser_seed;   /* 0x8F3F73B5CF1C9ADE for seqan3::views::minimiser_hash defaults */
    uint64_t bins;             /* B  = InputSummary::num_bins */
    uint64_t technical_bins;   /* TB = 64 * ceil(B/64) */
    uint64_t bin_size;         /* S  = rows */
    uint64_t hash_shift;       /* countl_zero(S) */
    uint64_t bin_words;        /* W  = TB / 64 (1..4) */
    uint8_t bin_to_category[256]; /* category INDEX of each user bin */
    uint64_t row_begin;        /* shard: this object holds rows [row_begin, row_end); 0,0 = all rows */
    uint64_t row_end;
} chn_index_desc;

int chn_index_create(const chn_index_desc *desc, chn_index **out);
/* Copy `n_rows` rows (n_rows * bin_words words) starting at global row `row_begin` from HOST memory.
 * Replaces the archive(ibf_) step of Index::serialize (include/index.hpp:130) after EF decoding; the
 * loader streams row blocks so host RAM never holds the whole plain index. */
int chn_index_upload_rows(chn_index *idx, uint64_t row_begin, uint64_t n_rows, const uint64_t *host_words);
/* Elias-Fano (sdsl::sd_vector) decode ON THE DEVICE: the alternative to chn_index_upload_rows for the loader, replacing the
 * archive(ibf_) step of Index::serialize (include/index.hpp:130) for the compressed IBF the file really holds.
 * `high` is a slice of m_high starting at bit `high_bit0` (a multiple of 64) with `n_high_words` words; `ones_before` = number of
 * set bits of m_high before that slice (the caller keeps the running popcount); `low` holds the packed m_low elements starting
 * with element `low_elem0` <= ones_before, `n_low_words` words (the last partial word included).  Every one of the slice is turned
 * into its plain bit position ((zeros before it) << wl | low part) and set in the index words; positions outside this shard's
 * rows are skipped.
 * The slice's work is QUEUED: the call returns as soon as `high` and `low` have been read (page-locked memory: chn_host_alloc; the copy
 * of pageable memory is what it waits for otherwise), so the caller prepares the next slice while the device decodes this one.
 * A closing call with n_high_words == 0 waits for every slice in flight and sets *bad_bits to the number of positions >= m_size or in
 * technical bins >= bins that the slices met (must be 0 for a well-formed file); calls with a slice set *bad_bits to 0.
 * chn_index_bin_popcounts and chn_stream_create wait for the slices as well. */
int chn_index_decode_ef(chn_index *idx, uint64_t m_size, uint32_t wl, const uint64_t *high, uint64_t high_bit0, uint64_t n_high_words,
                        uint64_t ones_before, const uint64_t *low, uint64_t low_elem0, uint64_t n_low_words, uint64_t *bad_bits);
/* set bits per technical bin over the rows this object holds (loader self-check iv); out[technical_bins] */
int chn_index_bin_popcounts(chn_index *idx, uint64_t *out);
/* Device pointer to the shard's words (for on-device index fabrication and for download in tests). */
int chn_index_device_words(chn_index *idx, uint64_t **device_words, uint64_t *n_words);
int chn_index_download_rows(chn_index *idx, uint64_t row_begin, uint64_t n_rows, uint64_t *host_words);
int chn_index_get_desc(const chn_index *idx, chn_index_desc *out);
int chn_index_destroy(chn_index *idx);

/* ---- model --------------------------------------------------------------------------------------------
 * Replaces: StatsModel + Model + KDEParams as read by ReadEntry::apply_model / call_host / call_category
 * (include/classify_stats.hpp:210-254,370-389,395-519; include/read_entry.hpp:157-279).
 * `paired` selects the caller: 0 = call_host (single-end dehost), 1 = call_category (paired dehost and every
 * `charon classify` run, include/read_entry.hpp:281-285). */
typedef struct chn_model {
    uint32_t struct_size;
    uint32_t num_categories;
    /* per category c: KDE datasets IN THE ORDER THE REFERENCE ITERATES THEM (default tables sorted by the
     * KDEParams constructor :214-218; trained tables in insertion order :234-240) */
    const float *const *pos_data; /* [C] host pointers */
    const uint32_t *pos_n;        /* [C] */
    const float *const *neg_data;
    const uint32_t *neg_n;
    float h_pos;                  /* 0.1   (:270) */
    float h_neg;                  /* 0.001 (:271) */
    float err_rate;               /* 300: stats::dexp(x, 300) (:371) */
    /* thresholds (include/dehost_arguments.hpp:30-37) */
    float min_quality;
    uint32_t min_length;
    float min_compression;
    int8_t confidence_threshold;  /* already narrowed to int8 (include/classify_stats.hpp:404,497) */
    uint8_t min_hits;             /* StatsModel::min_hits_ -- never initialised in the reference; caller's choice */
    uint8_t paired;               /* 1: call_category (what paired dehost runs, src/dehost_main.cpp:470) */
    uint8_t host_index;
    float confidence_probability_threshold;
    float host_unique_prop_lo_threshold;
    float min_proportion_difference;
    float min_prob_difference;
    /* Parametric models (`charon classify`, `charon dehost --dist gamma|beta`; include/classify_stats.hpp:116-208,289-339,
     * 377-381).  dist == CHN_DIST_KDE: the KDE datasets above are used and the two pointers are ignored.  Otherwise per
     * category three floats for the pos and three for the neg distribution: gamma (shape, loc, scale) -- reference defaults
     * pos {25, 0, 0.02}, neg {10, 0, 0.005} (:265-266) -- or beta (alpha, beta, unused) -- defaults pos {6, 4}, neg {6, 40}
     * (:267-268); the KDE datasets may then be NULL / empty. */
    uint32_t dist;
    const float *pos_params;      /* [C][3] */
    const float *neg_params;      /* [C][3] */
} chn_model;
#define CHN_DIST_KDE 0u
#define CHN_DIST_GAMMA 1u
#define CHN_DIST_BETA 2u

/* Fill `m` with the reference defaults (default KDE tables of src/dehost_main.cpp:23-206, sorted; thresholds
 * of include/dehost_arguments.hpp).  Pointers refer to static storage inside the library. */
int chn_model_default(chn_model *m, uint32_t num_categories, uint8_t host_index, int paired);

/* ---- streams and batches --------------------------------------------------------------------------------
 * One batch = what the reference calls a chunk (src/dehost_main.cpp:335-339), but sized for the GPU
 * (>= 64k reads rather than <= 255). */
typedef struct chn_stream_cfg {
    uint32_t struct_size;
    uint32_t flags;             /* CHN_STREAM_* */
    uint64_t max_reads;         /* per batch */
    uint64_t max_bases;         /* per batch, sum of padded segment lengths */
} chn_stream_cfg;
#define CHN_STREAM_PROFILE 1u   /* bracket every kernel with HIP events (chn_stream_profile) */
#define CHN_STREAM_TINY_LOG 2u  /* testing only: start with a deliberately undersized row log so that every batch takes the
                                 * overflow -> worst-case re-run path of chn_batch_wait */
/* A single-end read of 32 768 bases or more is rolled by the 64 lanes of one wavefront at once, in pieces that overlap by w - 1
 * bases (exact: a piece starts only where the window minimum is unique); shorter reads take one lane each.  Testing only:
 * CHN_STREAM_SPLIT_BUCKET(b) moves that limit to length class b (8 classes per octave: class = 8 (floor(log2 L) - 2) + the three
 * bits below L's leading one; 64 = 1 024 bases, the smallest accepted; 104 = the default; 255 = never split). */
#define CHN_STREAM_SPLIT_BUCKET(b) (((uint32_t)(b) & 0xffu) << 8)

int chn_stream_create(chn_index *idx, const chn_stream_cfg *cfg, chn_stream **out);
int chn_stream_destroy(chn_stream *s);
int chn_model_set(chn_stream *s, const chn_model *m);

/* Input batch.  Bases are 2-bit codes A0 C1 G2 T3, 4 per byte, least-significant bits first; base j of the
 * batch is bits [2*(j%16), +2) of little-endian dword j/16.  `nmask` (optional) has one bit per base in the
 * same order (bit j%32 of dword j/32): 1 = the base is N (any non-ACGT IUPAC letter, seqan3 dna5 rank 3);
 * its 2-bit code is then ignored.  Every segment must start at a multiple of 64 bases.
 * A read has one segment (single-end) or two (paired: mates are minimised separately and share one
 * accumulator, src/dehost_main.cpp:458-465).  mean_quality / compression are the host-side columns of
 * src/dehost_main.cpp:355-363 that gate the call (include/read_entry.hpp:242-252); NULL = 0.
 * If `on_device` is non-zero every pointer is a device pointer valid on the stream's device and no copy is
 * made (the caller keeps them alive until chn_batch_wait returns). */
typedef struct chn_batch {
    uint32_t struct_size;
    uint32_t on_device;
    uint64_t n_reads;
    uint64_t n_bases;            /* extent of `bases2`/`nmask` in bases (multiple of 64) */
    const uint32_t *bases2;
    const uint32_t *nmask;       /* may be NULL */
    const uint64_t *seg1_offset; /* [n] in bases */
    const uint32_t *seg1_length; /* [n] */
    const uint64_t *seg2_offset; /* [n] or NULL (single-end) */
    const uint32_t *seg2_length; /* [n] or NULL */
    const float *mean_quality;   /* [n] or NULL */
    const float *compression;    /* [n] or NULL */
    /* Deflate tallies for the `compression` column (get_compression_ratio, src/utils.cpp:114-124) ON THE DEVICE: non-zero = the
     * longest read (both mates together) to handle there, at most CHN_GZIP_MAX_LEN.  For every read the library then runs zlib's
     * level-6 deflate_slow as a kernel and returns the literal/length and distance code frequencies of its deflate block
     * (chn_result.gzip_tallies) -- the caller turns them into the exact gzip size with _tr_flush_block's arithmetic (a few
     * microseconds per read; charon_amd/csrc/host/gzip_size.hpp).  With tallies requested `compression` is not known when the
     * call kernel runs: its `compression < min_compression` gate (include/read_entry.hpp:189-191,250-252) is left open and the
     * CALLER must apply it (call = CHN_NO_CALL where the ratio is below the threshold). */
    uint32_t gzip_tallies;
    /* What comes back for the reads tallied on the device: CHN_GZIP_TALLIES (0) the tallies; CHN_GZIP_SIZES the gzip member
     * SIZES (chn_result.gzip_sizes) -- _tr_flush_block's tree arithmetic then runs on the device too (k_gzip_size) and only
     * four bytes per read are downloaded; CHN_GZIP_BOTH both. */
    uint32_t gzip_output;
} chn_batch;
#define CHN_GZIP_TALLIES 0u
#define CHN_GZIP_SIZES 1u
#define CHN_GZIP_BOTH 2u
#define CHN_GZIP_MAX_LEN 61440u  /* one wavefront holds the whole read in LDS; short reads run many wavefronts per CU, a 60 kb read one */
#define CHN_GZIP_TALLY_WORDS 320u /* per read: [0,286) literal/length code frequencies, [286,316) distance code frequencies,
                                   * [316] status: 0 = tallies valid, non-zero = not handled on the device (longer than asked for,
                                   * more than one deflate block): size this read on the host.
                                   * Long reads run few wavefronts per CU (one beyond 31 k letters): a caller with many idle host threads may prefer to keep
                                   * reads beyond ~16 k letters for itself (gzip_tallies = 16384) */

/* Per-read results (what a post-processed + classified ReadEntry holds, include/read_entry.hpp:23-32):
 * num_hashes_, counts_[C], unique_counts_[C], probabilities_[C], call_, confidence_score_.
 * proportions are not returned: they are float(count)/float(num_hashes) (:140-150), recomputed by the caller.
 * `flags` bit 0: a probability comparison that decides `call` was closer than 2e-6 relative (or a probability sits at the
 * float underflow edge of the `prob == 0` test of call_category), so the host should re-evaluate that read with its own
 * libm (the device exp() may differ from glibc in the last ulp; 2e-4 for gamma / beta models, whose densities the device forms from a
 * double log-density while the reference evaluates them in float: device-resident probabilities then agree to ~1e-5 only).
 * chn_batch_wait does that itself for host batches with host result buffers -- for gamma / beta it re-evaluates every read in float --;
 * with on_device results the flags are only reported and NOTHING is re-evaluated: a caller that needs the reference's `call` on
 * flagged reads runs chn_classify_counts on their counts. */
typedef struct chn_result {
    uint32_t struct_size;
    uint32_t on_device;        /* 0: pointers below are host buffers to fill; 1: receive device pointers */
    uint32_t *num_hashes;      /* [n] */
    uint32_t *counts;          /* [n*C] */
    uint32_t *unique_counts;   /* [n*C] */
    double *probabilities;     /* [n*C] */
    uint8_t *call;             /* [n] (CHN_NO_CALL = unclassified) */
    uint8_t *confidence;       /* [n] */
    uint8_t *flags;            /* [n] */
    uint16_t *gzip_tallies;    /* [n][CHN_GZIP_TALLY_WORDS] when the batch asked for them (may be NULL otherwise) */
    uint32_t *gzip_sizes;      /* [n] bytes of the gzip member (get_compression_ratio's numerator) when the batch asked for sizes;
                                * 0 = not handled on the device (too long, more than one deflate block): size it on the host */
} chn_result;

/* Up to THREE batches may be in flight per stream.  Two (submit, submit, wait, submit, wait, ...) let batch i's count and
 * model+call kernels run on a side HIP stream under batch i+1's minimise+probe kernel; they then usually finish only when
 * that kernel does, so a caller with HOST buffers keeps three in flight (submit i+2 before waiting for i): the upload of batch
 * i+2 then runs under the probe kernel of batch i+1.  chn_batch_wait returns the OLDEST batch in flight.  With on_device
 * results the returned pointers stay valid until the third-next submit.  Host results are downloaded in stream into
 * page-locked staging right behind the kernels that produce them; chn_batch_wait copies from there.
 * Scratch: the per-batch row log is sized for twice the minimiser density of random sequence (not for the worst case
 * of one minimiser per base); a batch that overruns it is detected on the device and re-run by chn_batch_wait on
 * worst-case buffers allocated at that point (CHN_E_NOMEM if they do not fit) -- results are identical either way.
 * A device batch (on_device != 0) whose segments are misaligned or reach beyond n_bases makes chn_batch_wait fail with
 * CHN_E_INVALID (such segments are never read). */
int chn_batch_submit(chn_stream *s, const chn_batch *b);   /* asynchronous */
int chn_batch_wait(chn_stream *s, chn_result *r);          /* blocks; fills / points `r` */
int chn_stream_sync(chn_stream *s);

/* Model + call only (k_model_call) on per-read counts the caller already holds -- used for reads that the
 * Result state machine cached while the KDE models were still training (include/result.hpp:139-151,181-198)
 * and that must be classified with the models as they are later.  All pointers are HOST arrays; outputs as in
 * chn_result.  Replaces ReadEntry::dehost / ReadEntry::classify (include/read_entry.hpp:281-291). */
int chn_classify_counts(chn_stream *s, uint64_t n_reads, const uint32_t *num_hashes, const uint32_t *counts,
                        const uint32_t *unique_counts, const uint32_t *lengths, const float *mean_quality,
                        const float *compression, double *probabilities, uint8_t *call, uint8_t *confidence);

/* ---- row-sharded ("hash-bin" sharded) mode, dense exchange (the checker of the sparse exchange below) -----------
 * For an index too large for one GPU: rank r creates a chn_index with row_begin/row_end = its slice and a stream on it.
 * Per batch, on EVERY rank and for the SAME batch:
 *   1. chn_shard_minimise   all reads are minimised (redundantly); returns E = number of minimisers of the batch
 *   2. chn_shard_probe      partial[e][i][w] = word w of row hash_i(minimiser e) if this rank owns the row, else 0
 *   3. the caller sums `partial` (E*h*W uint64) over ranks with ONE all-reduce (RCCL; sum == select because exactly
 *      one rank owns each row) -- the library itself has no RCCL dependency
 *   4. chn_shard_finish     AND over the h hash functions, counts, model+call; then chn_batch_wait as usual.
 * `dev_partial` is a device buffer of the caller (e.g. a torch tensor). No reference counterpart (the reference is
 * single-process); replaces the same loop body as chn_batch_submit. */
int chn_shard_minimise(chn_stream *s, const chn_batch *b, uint64_t *n_entries);
int chn_shard_probe(chn_stream *s, const chn_index *shard, uint64_t *dev_partial, uint64_t capacity_words);
int chn_shard_finish(chn_stream *s, const uint64_t *dev_partial);

/* ---- row-sharded mode, sparse exchange -----------------------------------------------------------------------
 * The faster way to use an index whose rows are spread over n_ranks GPUs (rank r holds rows [row_splits[r], row_splits[r+1])):
 * every rank classifies ITS OWN reads, asks the owners for the rows it needs and gets them back.  Per batch, on every rank:
 *   1. chn_shardx_minimise   (asynchronous) the rank's reads are minimised; nothing is probed yet
 *   2. chn_shardx_counts     (blocks) send_counts[r] = number of probes (minimiser x hash function) whose row rank r owns,
 *                            *n_probes = their sum
 *   3. chn_shardx_queries    (asynchronous) dev_queries[n_probes]: the owner-local row number of every probe, grouped by owner
 *                            rank in rank order
 *   4. the caller exchanges the groups (all-to-all, 4 bytes per probe; RCCL on torch tensors -- the library has no RCCL dependency)
 *   5. chn_shardx_serve      (asynchronous; on the OWNER) dev_rows_out[j][w] = word w of the shard's row dev_queries_in[j]
 *   6. the caller sends the rows back the way the queries came (second all-to-all, 8 * bin_words bytes per probe), so that
 *      dev_rows_back is laid out exactly like dev_queries
 *   7. chn_shardx_finish     (asynchronous) AND over the h hash functions, counts, model+call; then chn_batch_wait as usual.
 * The asynchronous calls run on the stream's HIP stream: call chn_stream_sync before handing a buffer to a collective that runs
 * on another stream.  The stream `s` may be created on any index object of the same IBF (its rows are not read by steps 1-4, 7);
 * one sharded batch at a time per stream (use two streams to overlap step 1 of the next batch with step 5 of this one).
 * No reference counterpart (the reference is single-process); replaces the same loop body as chn_batch_submit. */
int chn_shardx_minimise(chn_stream *s, const chn_batch *b);
int chn_shardx_counts(chn_stream *s, uint32_t n_ranks, const uint64_t *row_splits /* [n_ranks + 1] */, uint64_t *n_probes,
                      uint64_t *send_counts /* [n_ranks] */);
int chn_shardx_queries(chn_stream *s, uint32_t *dev_queries, uint64_t capacity);
int chn_shardx_serve(chn_stream *s, const chn_index *shard, const uint32_t *dev_queries_in, uint64_t n_in, uint64_t *dev_rows_out);
int chn_shardx_finish(chn_stream *s, const uint64_t *dev_rows_back);

/* ---- index construction (`charon index`, src/index_main.cpp:118-160,238-263) ------------------------------
 * chn_minimisers: all minimisers emitted for the segments of `b` (seqan3 minimiser_hash with the stream's k, w), with
 * repeats, in an unspecified but deterministic order, copied to host memory.  Replaces the per-record
 * `record.sequence() | hash_adaptor` of count_and_store_hashes (:142-148); segments may overlap in `bases2`, which is how
 * long reference sequences are cut into chunks overlapping by w-1 bases (the union over such chunks is exactly the set
 * of window minima of the whole sequence).
 * chn_index_emplace: ibf.emplace(value, bin) for every value (:252-255); values are host memory. */
int chn_minimisers(chn_stream *s, const chn_batch *b, uint64_t *host_values, uint64_t capacity, uint64_t *n_values);
int chn_index_emplace(chn_index *idx, const uint64_t *host_values, uint64_t n_values, uint32_t bin);

/* Per-kernel device time accumulated since the last reset (CHN_STREAM_PROFILE streams only).
 * which: 0 = minimise+probe kernel, 1 = count kernel, 2 = model+call kernel, 3 = whole batch chain;
 * 4 (any stream): *launches = number of batches chn_batch_wait re-ran on worst-case buffers after a row-log overflow;
 * 5 (any stream): *launches = row fetches the last waited batch's minimise+probe kernel issued (h per minimiser; fewer for an index of
 *   at most four bins, whose rows are fetched one at a time and only while the AND so far still has a bin set). */
int chn_stream_profile(chn_stream *s, int which, double *total_ms, uint64_t *launches, int reset);
/* Algorithmic bytes of the last batch by SURVEY 8(d): sum over reads of ceil(L/4) + M*h*W*8 + (8 + 8C). */
int chn_stream_last_batch_bytes(chn_stream *s, uint64_t *bytes, uint64_t *total_minimisers);

/* ---- synthetic workload fabrication on the device (bench / tests; no reference counterpart) ---------- */
/* Measurement aid: the rate this device sustains for NOTHING BUT the index's probe pattern -- independent uniformly random row
 * fetches of 8 * bin_words bytes from THIS index's words (one load per thread in flight, 32 wavefronts per CU, `nt` cache policy if
 * nt != 0) -- in row fetches per second.  bench.py reports k_minimise_probe's probe rate against it (roofline.gather_roof). */
int chn_index_gather_roof(chn_index *idx, int nt, double *fetches_per_s);
/* Random 2-bit genomes: n_genomes x genome_len bases (genome_len multiple of 64), counter-based PRNG. */
int chn_synth_genomes(int device, uint64_t seed, uint64_t n_genomes, uint64_t genome_len, uint32_t **dev_bases2);
/* Set every bit of user bins [0,B) of every row with probability `density` (background fill). */
int chn_synth_fill_index(chn_index *idx, uint64_t seed, double density);
/* Insert the minimisers of genome g into bin genome_bin[g] (IBF emplace, 3 rows each). */
int chn_synth_plant(chn_index *idx, const uint32_t *dev_bases2, uint64_t n_genomes, uint64_t genome_len,
                    const uint8_t *genome_bin /* host, [n_genomes] */);
/* Sample `n_reads` reads of `read_len` bases: read i comes from genome (hash % n_genomes) with probability
 * 1 - random_fraction (uniform start, i.i.d. substitutions at `sub_rate`), else is uniformly random.
 * Writes a packed batch into freshly allocated device buffers (segments padded to 64 bases). */
typedef struct chn_synth_reads_out {
    uint32_t *bases2;
    uint64_t *seg1_offset;
    uint32_t *seg1_length;
    float *mean_quality;
    float *compression;
    uint64_t n_bases;
} chn_synth_reads_out;
/* `first_read_id`: global index of read 0 of this batch; a read's content depends only on (seed, global index), so
 * the union of the batches of N ranks equals one batch of N times the size. */
int chn_synth_reads(int device, uint64_t seed, const uint32_t *dev_genomes, uint64_t n_genomes, uint64_t genome_len,
                    uint64_t first_read_id, uint64_t n_reads, uint32_t read_len_min, uint32_t read_len_max, double sub_rate,
                    double random_fraction, float mean_quality, chn_synth_reads_out *out);
/* Page-locked host memory.  Host batches whose arrays live in such memory are uploaded asynchronously on a copy stream,
 * so with batches in flight the upload of the next one overlaps the kernels of those before; pageable memory works too but
 * its copies are staged synchronously by the runtime. */
int chn_host_alloc(uint64_t bytes, void **ptr);
int chn_host_free(void *ptr);
int chn_device_malloc(int device, uint64_t bytes, void **ptr);
int chn_device_free(int device, void *ptr);
int chn_device_upload(int device, void *dev_dst, const void *host_src, uint64_t bytes);
int chn_device_download(int device, void *host_dst, const void *dev_src, uint64_t bytes);

const char *chn_last_error(void);
const char *chn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CHARON_HIP_H */
