// ORACLE -- TEST INFRASTRUCTURE ONLY (see charon_oracle.hpp).  C entry points for ctypes so that
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive the CPU restatement.
#include "charon_oracle.hpp"

#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace oracle;

extern "C" {

int orc_load_tables(const char *path) {
    try { load_default_tables(path); return 0; } catch (std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return -1; }
}

// ---- primitives -------------------------------------------------------------------------------
// sigma = 5: chars are folded with dna5_rank; sigma = 4: A0 C1 G2 T3 (documentation vectors)
static std::vector<uint8_t> ranks_sigma(const char *seq, uint64_t n, int sigma) {
    std::vector<uint8_t> r(n);
    for (uint64_t i = 0; i < n; ++i) {
        if (sigma == 5) r[i] = dna5_rank(seq[i]);
        else { switch (seq[i]) { case 'A': r[i] = 0; break; case 'C': r[i] = 1; break; case 'G': r[i] = 2; break; default: r[i] = 3; } }
    }
    return r;
}
uint64_t orc_kmer_hashes(const char *seq, uint64_t n, int k, int sigma, uint64_t *out, uint64_t cap) {
    auto v = kmer_hashes(ranks_sigma(seq, n, sigma), (unsigned)k, sigma == 5 ? alphabet_dna5() : alphabet_dna4());
    for (uint64_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}
uint64_t orc_minimisers(const char *seq, uint64_t n, int k, int w, int sigma, uint64_t *out, uint64_t cap) {
    auto v = minimiser_hash(ranks_sigma(seq, n, sigma), (unsigned)k, (unsigned)w, sigma == 5 ? alphabet_dna5() : alphabet_dna4());
    for (uint64_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}
uint64_t orc_hash_and_fit(uint64_t x, int seed_idx, uint64_t bin_size) {
    return hash_and_fit_row(x, IBF_SEEDS[seed_idx], bin_size, clz64(bin_size));
}
uint64_t orc_bin_size_in_bits(uint64_t n, int h, double fpr) { return bin_size_in_bits(n, (unsigned)h, fpr); }
float orc_compression_ratio(const char *seq, uint64_t n) { return get_compression_ratio(std::string(seq, n)); }
float orc_dexp300(float x) { return dexp300(x); }
// Model::prob for the default model (which: 0 -> pos, 1 -> neg, 2 -> p_pos, 3 -> p_neg, 4 -> p_err)
double orc_default_model_prob(float x, int which) {
    Model m;
    if (which == 0) return m.prob(x).pos;
    if (which == 1) return m.prob(x).neg;
    if (which == 2) return m.k_pos.prob(x);
    if (which == 3) return m.k_neg.prob(x);
    return dexp300(x);
}

// ---- index ------------------------------------------------------------------------------------
void *orc_index_new(int k, int w, uint64_t nbins, uint64_t bin_size, int nhash, int ncat, const uint8_t *bin_to_cat,
                    const char *const *cat_names) {
    Index *idx = new Index();
    idx->kmer_size = (uint8_t)k; idx->window_size = (uint8_t)w;
    for (int c = 0; c < ncat; ++c) idx->summary.categories.push_back(cat_names[c]);
    idx->summary.num_bins = (uint8_t)nbins;
    for (uint64_t b = 0; b < nbins; ++b) {
        idx->summary.bin_to_category[(uint8_t)b] = cat_names[bin_to_cat[b]];
        idx->summary.filepath_to_bin.emplace_back("synthetic_bin_" + std::to_string(b) + ".fa", (uint8_t)b);
        idx->stats.records_per_bin[(uint8_t)b] = 0;
        idx->stats.hashes_per_bin[(uint8_t)b] = 0;
    }
    idx->stats.num_files = (uint32_t)nbins;
    idx->init_ibf(nbins, bin_size, (uint64_t)nhash);
    return idx;
}
void *orc_index_build_from_fasta(int nfiles, const char *const *paths, const char *const *cats, int ncat,
                                 const char *const *cat_order, int k, int w, uint64_t force_bin_size) {
    try {
        std::vector<std::pair<std::string, std::string>> fc;
        for (int i = 0; i < nfiles; ++i) fc.emplace_back(paths[i], cats[i]);
        std::vector<std::string> order;
        for (int i = 0; i < ncat; ++i) order.push_back(cat_order[i]);
        return new Index(build_index(fc, order, (unsigned)w, (unsigned)k, 3, 0.01, force_bin_size));
    } catch (std::exception &e) { std::fprintf(stderr, "orc_index_build_from_fasta: %s\n", e.what()); return nullptr; }
}
void orc_index_free(void *h) { delete (Index *)h; }
uint64_t *orc_index_words(void *h) { return ((Index *)h)->plain.data(); }
uint64_t orc_index_nwords(void *h) { return ((Index *)h)->plain.size(); }
void orc_index_emplace(void *h, uint64_t value, uint64_t bin) { ((Index *)h)->emplace(value, bin); }
void orc_index_emplace_many(void *h, const uint64_t *values, uint64_t n, uint64_t bin) {
    Index *idx = (Index *)h;
    for (uint64_t i = 0; i < n; ++i) idx->emplace(values[i], bin);
}
void orc_index_compress(void *h) { ((Index *)h)->compress(); }
void orc_index_use_ef(void *h, int on) { ((Index *)h)->use_ef = on != 0; }
int orc_index_store(void *h, const char *path) {
    try { store_index(path, *(Index *)h); return 0; } catch (std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return -1; }
}
void *orc_index_load(const char *path) {
    try { Index *idx = new Index(); load_index(*idx, path); return idx; }
    catch (std::exception &e) { std::fprintf(stderr, "orc_index_load: %s\n", e.what()); return nullptr; }
}
// params: [k, w, bins, technical_bins, bin_size, hash_shift, bin_words, hash_funs, ncat, host_index, ef_ones, ef_wl]
void orc_index_params(void *h, uint64_t *out) {
    Index *i = (Index *)h;
    out[0] = i->kmer_size; out[1] = i->window_size; out[2] = i->bins; out[3] = i->technical_bins; out[4] = i->bin_size;
    out[5] = i->hash_shift; out[6] = i->bin_words; out[7] = i->hash_funs; out[8] = i->summary.num_categories();
    out[9] = i->get_host_index(); out[10] = i->ef.ones; out[11] = i->ef.wl;
}
void orc_index_bin_to_cat(void *h, uint8_t *out) {
    Index *i = (Index *)h;
    for (unsigned b = 0; b < i->summary.num_bins; ++b) out[b] = i->summary.category_index(i->summary.bin_to_category.at((uint8_t)b));
}
int orc_index_category_name(void *h, int c, char *buf, int cap) {
    Index *i = (Index *)h;
    std::snprintf(buf, (size_t)cap, "%s", i->summary.categories.at((size_t)c).c_str());
    return 0;
}
void orc_index_bulk_contains(void *h, uint64_t value, uint64_t *out) { ((Index *)h)->bulk_contains(value, out); }
uint64_t orc_sd_get_int(void *h, uint64_t bit) { return ((Index *)h)->ef.get_int(bit); }
// select_support_mcl blocks of a loaded file: 1 if present; orc_sd_select answers select_b(i), i from 1, from the stored blocks
int orc_sd_has_select(void *h) { return ((Index *)h)->ef.has_sel ? 1 : 0; }
uint64_t orc_sd_select_args(void *h, int b) { const SdVector &e = ((Index *)h)->ef; return b ? e.sel1.m_arg_cnt : e.sel0.m_arg_cnt; }
uint64_t orc_sd_select(void *h, int b, uint64_t i) { const SdVector &e = ((Index *)h)->ef; return b ? e.sel1.select(i) : e.sel0.select(i); }
// the two serialised select_support_mcl blocks (ones, then zeros) of a raw bit vector, as store_index appends them
int orc_select_blocks(const uint64_t *words, uint64_t nbits, const char *path) {
    try {
        std::ofstream os(path, std::ios::binary);
        BinWriter w(os);
        for (int b = 1; b >= 0; --b) { SelectMcl s; s.init(words, nbits, b); w.select_mcl(s); }
        return os ? 0 : -1;
    } catch (std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return -1; }
}

// ---- the per-read path (A4-A10) on in-memory reads -------------------------------------------
// seqs: concatenated ASCII bases; offsets[n+1]; mate_split[i] (or NULL) = length of mate 1 of read i
// (mate 2 = the rest); quals: optional concatenated phred+33 with the same offsets (NULL -> mean_quality = mq_const).
// Outputs (all caller allocated): num_hashes[n], counts[n*C], unique[n*C], props[n*C], uprops[n*C],
// probs[n*C] (double), call[n], conf[n], mean_q[n], compression[n].
// Model: the default KDE (no training), thresholds from `thr` (see below) -- i.e. what every read but the
// dropped first one gets in a no-extract run.
struct OrcThresholds {
    float min_quality; uint32_t min_length; float min_compression; uint8_t confidence_threshold;
    float confidence_probability_threshold, host_unique_prop_lo_threshold, min_proportion_difference, min_prob_difference;
    uint8_t min_hits; uint8_t paired; uint8_t with_gzip; uint8_t dist;  // dist: 0 kde, 1 gamma, 2 beta (default parameters)
};
double orc_process_reads(void *h, const char *seqs, const uint64_t *offsets, uint64_t n, const uint32_t *mate_split,
                         const char *quals, float mq_const, const OrcThresholds *thr, int threads, uint32_t *num_hashes,
                         uint32_t *counts, uint32_t *unique, float *props, float *uprops, double *probs, uint8_t *call,
                         uint8_t *conf, float *mean_q, float *compression) {
    Index *idx = (Index *)h;
    DehostArguments opt;
    opt.min_quality = thr->min_quality; opt.min_length = thr->min_length; opt.min_compression = thr->min_compression;
    opt.confidence_threshold = thr->confidence_threshold;
    opt.confidence_probability_threshold = thr->confidence_probability_threshold;
    opt.host_unique_prop_lo_threshold = thr->host_unique_prop_lo_threshold;
    opt.min_proportion_difference = thr->min_proportion_difference; opt.min_prob_difference = thr->min_prob_difference;
    opt.min_hits = thr->min_hits;
    opt.dist = thr->dist == 1 ? "gamma" : thr->dist == 2 ? "beta" : "kde";
    StatsModel model(opt, idx->summary);
    const unsigned C = idx->summary.num_categories();
    const uint8_t host = idx->summary.host_category_index();
    auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 16)
    for (long i = 0; i < (long)n; ++i) {
        Record r1, r2;
        uint64_t b = offsets[i], e = offsets[i + 1];
        uint64_t s = mate_split ? (uint64_t)mate_split[i] : (e - b);
        r1.seq.assign(seqs + b, seqs + b + s);
        r1.id = "r";
        if (quals) r1.qual.assign(quals + b, quals + b + s);
        if (mate_split) { r2.seq.assign(seqs + b + s, seqs + e); if (quals) r2.qual.assign(quals + b + s, quals + e); }
        ReadEntry en = process_read(*idx, idx->summary, r1, mate_split ? &r2 : nullptr, !thr->with_gzip);
        if (!quals) en.mean_quality_ = mq_const;
        if (thr->paired) en.classify(model); else en.dehost(model, host);
        if (num_hashes) num_hashes[i] = en.num_hashes_;
        for (unsigned c = 0; c < C; ++c) {
            if (counts) counts[i * C + c] = en.counts_[c];
            if (unique) unique[i * C + c] = en.unique_counts_[c];
            if (props) props[i * C + c] = en.proportions_[c];
            if (uprops) uprops[i * C + c] = en.unique_proportions_[c];
            if (probs) probs[i * C + c] = en.probabilities_[c];
        }
        if (call) call[i] = en.call_;
        if (conf) conf[i] = en.confidence_score_;
        if (mean_q) mean_q[i] = en.mean_quality_;
        if (compression) compression[i] = en.compression_;
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// full `charon dehost` on files -> TSV text into a caller buffer (returns bytes needed)
uint64_t orc_dehost_files_dist(void *h, const char *reads1, const char *reads2, int run_extract, int chunk_size, int threads,
                               int num_reads_to_fit, float min_quality, int confidence, const char *dist, char *out, uint64_t cap);
uint64_t orc_dehost_files(void *h, const char *reads1, const char *reads2, int run_extract, int chunk_size, int threads,
                          int num_reads_to_fit, float min_quality, int confidence, char *out, uint64_t cap) {
    return orc_dehost_files_dist(h, reads1, reads2, run_extract, chunk_size, threads, num_reads_to_fit, min_quality, confidence, "kde", out, cap);
}
uint64_t orc_dehost_files_dist(void *h, const char *reads1, const char *reads2, int run_extract, int chunk_size, int threads,
                               int num_reads_to_fit, float min_quality, int confidence, const char *dist, char *out, uint64_t cap) {
    Index *idx = (Index *)h;
    DehostArguments opt;
    if (dist && dist[0]) opt.dist = dist;
    opt.read_file = reads1;
    if (reads2 && reads2[0]) { opt.read_file2 = reads2; opt.is_paired = true; opt.min_length = 80; }
    opt.run_extract = run_extract != 0;
    opt.chunk_size = (uint8_t)chunk_size; opt.threads = (uint8_t)threads; opt.num_reads_to_fit = (uint16_t)num_reads_to_fit;
    opt.min_quality = min_quality; opt.confidence_threshold = (uint8_t)confidence;
    std::ostringstream os;
    try { dehost_run(opt, *idx, os); } catch (std::exception &e) { std::fprintf(stderr, "orc_dehost_files: %s\n", e.what()); return 0; }
    std::string s = os.str();
    if (out && cap) std::memcpy(out, s.data(), std::min<uint64_t>(cap, s.size()));
    return s.size();
}

// full `charon classify` on files -> TSV text (classify defaults of include/classify_arguments.hpp; dist "beta" / "gamma")
uint64_t orc_classify_files(void *h, const char *reads1, const char *reads2, int run_extract, int chunk_size, int threads,
                            int num_reads_to_fit, const char *dist, char *out, uint64_t cap) {
    Index *idx = (Index *)h;
    DehostArguments opt = classify_defaults();
    opt.read_file = reads1;
    if (reads2 && reads2[0]) { opt.read_file2 = reads2; opt.is_paired = true; opt.min_length = 80; }
    opt.run_extract = run_extract != 0;
    opt.chunk_size = (uint8_t)chunk_size; opt.threads = (uint8_t)threads; opt.num_reads_to_fit = (uint16_t)num_reads_to_fit;
    if (dist && dist[0]) opt.dist = dist;
    std::ostringstream os;
    try { dehost_run(opt, *idx, os); } catch (std::exception &e) { std::fprintf(stderr, "orc_classify_files: %s\n", e.what()); return 0; }
    std::string s = os.str();
    if (out && cap) std::memcpy(out, s.data(), std::min<uint64_t>(cap, s.size()));
    return s.size();
}
// the two statslib densities, for the scipy cross-check: kind 1 = dgamma(x, shape p0, scale p1), 2 = dbeta(x, alpha p0, beta p1)
float orc_density(int kind, float x, float p0, float p1) { return kind == 1 ? stats_dgamma(x, p0, p1) : stats_dbeta(x, p0, p1); }
// GammaParams::fit / fit_loc and BetaParams::fit on caller data: out = {shape, loc, scale} or {alpha, beta, 0}
void orc_fit(int kind, const float *data, uint64_t n, int loc_only, float *out) {
    std::vector<float> v(data, data + n);
    if (kind == 1) { GammaParams g{out[0], out[1], out[2]}; if (loc_only) g.fit_loc(v); else g.fit(v); out[0] = g.shape; out[1] = g.loc; out[2] = g.scale; }
    else { BetaParams b{out[0], out[1], 0}; b.fit(v); out[0] = b.alpha; out[1] = b.beta; out[2] = 0; }
}

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
