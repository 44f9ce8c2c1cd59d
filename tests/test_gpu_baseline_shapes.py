"""GPU parity tests (-m gpu) on the FULL-SIZE shapes of BASELINE.json's single-GPU configurations: the fused path of
configs[1] (2-category 1 GiB index, 1 M x 5 kb reads) and the general path of configs[2] (39 GB stand-in index with more
than 2^31 rows -> 64-bit row offsets, W = 2, non-temporal probes; mixed-length reads 500 b - 50 kb).  Index and reads are
fabricated on the device exactly as bench.py does; a sample of the reads (always including the longest ones) is replayed
through the CPU oracle on index rows downloaded from HBM, and the size-independent split-invariance property is checked
on the whole batch (one batch == the same reads in four batches: different wavefront grouping, different log layout)."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import charon_amd.api as api
    return api


def _classify(api, g, gen, n_gen, glen, first, count, lmin, lmax, ncat, keep_sample=None, two_in_flight=False):
    rd = api.synth_reads(0, 42, gen, n_gen, glen, count, lmin, lmax, 0.05, 0.1, 40.0, first_read_id=first)
    st = api.Stream(g, count, rd.n_bases)
    st.set_model(api.default_model(ncat, 0))
    st.submit_device(count, rd.n_bases, rd.bases2, rd.seg1_offset, rd.seg1_length, rd.mean_quality, rd.compression)
    if two_in_flight:  # the same batch again behind it: batch 1's count/model kernels overlap batch 2's minimise+probe
        st.submit_device(count, rd.n_bases, rd.bases2, rd.seg1_offset, rd.seg1_length, rd.mean_quality, rd.compression)
    out = util.download_results(api, st.wait_device(), count, ncat)
    if two_in_flight:
        util.assert_same_results(out, util.download_results(api, st.wait_device(), count, ncat))
    assert st.profile(4)[1] == 0  # no row-log overflow on ordinary sequence
    sample = None
    if keep_sample is not None:
        from charon_amd import pack
        lens = api.device_download(0, rd.seg1_length, count * 4, np.uint32)
        offs = api.device_download(0, rd.seg1_offset, count * 8, np.uint64)
        pick = keep_sample(lens)
        seqs = []
        for i in pick:  # one read at a time: the sample is spread over a multi-GB base buffer
            o, l = int(offs[i]), int(lens[i])
            nb = (l + 63) // 64 * 64
            words = api.device_download(0, rd.bases2 + o // 4, max(nb // 4, 16), np.uint32)
            seqs.append(pack.unpack_reads(words, [0], [l])[0])
        sample = (np.asarray(pick), seqs)
    st.destroy()
    util.free_synth_reads(api, rd)
    return out, sample


def _oracle_index(api, oracle_lib, g, B, S, b2c, cats):
    oidx = oracle_lib.Index.new(B, S, b2c, cats)
    words = oidx.words()
    Wd = g.desc.bin_words
    chunk_rows = max(1, (1 << 30) // (8 * Wd))
    for r0 in range(0, S, chunk_rows):
        nr = min(chunk_rows, S - r0)
        assert api.lib().chn_index_download_rows(g.h, r0, nr, words[r0 * Wd:].ctypes.data) == 0
    return oidx


def test_config2_full_size_fused_shape(api, oracle_lib):
    """BASELINE configs[1]: B = 2 (one bin per category -> FUSED counters), W = 1, S = 2^27 (1 GiB), 1 M x 5 kb reads."""
    B, S, n, L, glen = 2, 1 << 27, 1 << 20, 5000, 1 << 24
    g = api.Index(api.make_desc(B, S, [0, 1], 2, 0))
    gen = api.synth_genomes(0, 43, B, glen)
    g.synth_fill(43, 0.215)
    g.synth_plant(gen, B, glen, [0, 1])
    r = util.rng(5)
    whole, (pick, seqs) = _classify(api, g, gen, B, glen, 0, n, L, L, 2, two_in_flight=True,
                                    keep_sample=lambda lens: np.sort(r.choice(len(lens), 512, replace=False)))
    parts = [_classify(api, g, gen, B, glen, i * (n // 4), n // 4, L, L, 2)[0] for i in range(4)]
    util.assert_same_results(whole, {k: np.concatenate([p[k] for p in parts]) for k in whole})
    assert (whole["call"] == 0).sum() > n // 3 and (whole["call"] == 1).sum() > n // 3
    assert abs(whole["num_hashes"].mean() - 448.5) < 2
    oidx = _oracle_index(api, oracle_lib, g, B, S, [0, 1], ["human", "microbial"])
    cat, offs, _ = util.concat(seqs)
    util.assert_parity({k: v[pick] for k, v in whole.items() if k != "flags"}, oidx.process_reads(cat, offs))
    oidx.free()
    api.device_free(0, gen)
    g.destroy()


def test_config3_39gb_shape_mixed_length_reads(api, oracle_lib):
    """BASELINE configs[2]: B = 100 (50 human / 50 microbial bins), TB = 128, W = 2, S = 2 437 500 000 (> 2^31 rows: 39.0 GB
    of plain rows), reads log-uniform in 500 b - 50 kb.  65 536 reads (0.7 G bases) per batch."""
    B, S, n, glen = 100, 2437500000, 1 << 16, 1 << 20
    b2c = [b % 2 for b in range(B)]
    g = api.Index(api.make_desc(B, S, b2c, 2, 0))
    assert g.desc.bin_words == 2
    gen = api.synth_genomes(0, 43, B, glen)
    g.synth_fill(43, 0.215)
    g.synth_plant(gen, B, glen, list(range(B)))
    r = util.rng(6)

    def pick_sample(lens):  # the 64 longest, the 64 shortest and 512 random reads
        o = np.argsort(lens, kind="stable")
        return np.unique(np.concatenate([o[-64:], o[:64], r.choice(len(lens), 512, replace=False)]))
    whole, (pick, seqs) = _classify(api, g, gen, B, glen, 0, n, 500, 50000, 2, keep_sample=pick_sample, two_in_flight=True)
    lens = np.array([len(s) for s in seqs])
    assert lens.max() > 45000 and lens.min() < 520 and len(pick) >= 512
    parts = [_classify(api, g, gen, B, glen, i * (n // 4), n // 4, 500, 50000, 2)[0] for i in range(4)]
    util.assert_same_results(whole, {k: np.concatenate([p[k] for p in parts]) for k in whole})
    assert (whole["call"] == 0).sum() > n // 4 and (whole["call"] == 1).sum() > n // 4
    # rows above 2^31 are really probed: with 32-bit row arithmetic the planted genomes would not be found
    oidx = _oracle_index(api, oracle_lib, g, B, S, b2c, ["human", "microbial"])
    cat, offs, _ = util.concat(seqs)
    orc = oidx.process_reads(cat, offs)
    util.assert_parity({k: v[pick] for k, v in whole.items() if k != "flags"}, orc)
    assert orc["unique"].sum() > 0 and (orc["call"] != 255).sum() > len(pick) // 2
    oidx.free()
    api.device_free(0, gen)
    g.destroy()
