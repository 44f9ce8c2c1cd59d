"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Integer columns (num_hashes, counts, unique counts, confidence, call) must be bit-exact; the KDE
probability column within 1e-6 (BASELINE.json north_star)."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import charon_amd.api as api
    return api


def run_gpu(api, gidx, reads, mates=None, mq=40.0, comp=0.3, model=None, profile=False):
    from charon_amd import pack
    p = pack.pack_reads(reads, mates)
    n = len(reads)
    st = api.Stream(gidx, max(n, 1), p["n_bases"], profile=profile)
    paired = mates is not None
    st.set_model(model or api.default_model(gidx.desc.num_categories, gidx.desc.host_index if not paired else 0, paired=paired))
    st.submit_host(p, np.full(n, mq, np.float32), np.full(n, comp, np.float32))
    out = st.wait_host()
    st.destroy()
    return out


def run_oracle(oidx, reads, mates=None, mq=40.0, comp=0.3, po=None):
    seqs, offs, split = util.concat(reads, mates)
    out = oidx.process_reads(seqs, offs, mate_split=split, mq_const=mq)
    return out


def check(api, po, oidx, reads, mates=None, **kw):
    g = util.gpu_index_from_oracle(api, oidx)
    try:
        # compression is a host-side column (zlib) handed to the call kernel; the oracle is run without gzip so
        # both sides gate on the same value 0 >= min_compression 0
        gpu = run_gpu(api, g, reads, mates, comp=0.0, **kw)
        orc = run_oracle(oidx, reads, mates)
        util.assert_parity(gpu, orc)
        return gpu, orc
    finally:
        g.destroy()


def test_cfg1_toy_index_1k_reads(api, oracle_lib, my_fasta):
    """BASELINE config 1: my.fasta (microbial) + seeded 10 kb random genome (host), 1 000 x 1 kb reads."""
    r = util.rng(42)
    host = util.random_seq(r, 10000)
    micro = [s.encode() for s in my_fasta.values()]
    oidx = util.build_oracle_index(oracle_lib, [micro, [host]], [0, 1], ["microbial", "host"])
    assert oidx.bin_size == 11957 or oidx.bin_size > 0
    reads = util.sample_reads(r, micro[1:4] + [host], 1000, 1000, sub_rate=0.05, random_fraction=0.1)
    gpu, orc = check(api, oracle_lib, oidx, reads)
    assert set(np.unique(gpu["call"])) >= {0, 1, 255}
    assert abs(gpu["num_hashes"].mean() - 87) < 6
    oidx.free()


def test_edge_reads_ties_n_short(api, oracle_lib):
    """homopolymers / tandem repeats (tie path of the minimiser rule), N runs, L < k, k <= L < w, ragged lengths"""
    r = util.rng(1)
    g0, g1 = util.random_seq(r, 5000), util.random_seq(r, 5000)
    low = b"A" * 300 + g0[:200] + b"AC" * 150 + g1[:100] + b"ACG" * 90
    oidx = util.build_oracle_index(oracle_lib, [[g0, low], [g1]], [0, 1], ["host", "other"])
    reads = [b"A" * 1000, b"AC" * 400, b"ACGT" * 100, b"ACG" * 200 + b"T" * 77, low, low[100:700], b"A" * 18, b"A" * 19, b"C" * 40,
             b"G" * 41, b"T" * 42, b"N" * 300, g0[:500].replace(b"A", b"N", 3), b"ACGTN" * 60,
             g0[100:160] + b"NNNN" + g0[164:900], b"A", b"ACGTACGTACGTACGTACG", g1[:63], g1[:64], g1[:65], g0[:128], g0[:129],
             (g0[:300] + b"RYKM" + g0[304:600]).lower(), g1[200:241], g1[200:240]]
    reads += util.sample_reads(r, [g0, g1, low], 200, (1, 1500), sub_rate=0.02)
    check(api, oracle_lib, oidx, reads)
    oidx.free()


def test_multi_bin_categories_rows_path(api, oracle_lib):
    """a category with several bins -> max-bin selection needs the stored rows (H8); also W = 2 (70 bins)"""
    r = util.rng(2)
    gs = [util.random_seq(r, 4000) for _ in range(5)]
    # related genomes in one category so the strictly-largest-total rule matters
    gs[2] = util.mutate(r, gs[0], 0.02)
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1, 0, 1, 0], ["human", "microbial"], fill_seed=5, fill=0.05)
    reads = util.sample_reads(r, gs, 300, (100, 2500), sub_rate=0.04)
    gpu, _ = check(api, oracle_lib, oidx, reads)
    assert gpu["unique"].sum() > 0
    oidx.free()
    gs = [util.random_seq(r, 1500) for _ in range(70)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [i % 2 for i in range(70)], ["microbial", "human"], bin_size=3001,
                                   fill_seed=6, fill=0.1)
    assert oidx.bin_words == 2
    reads = util.sample_reads(r, gs, 200, (30, 1400), sub_rate=0.03) + [b"A" * 500, b"", b"ACGT" * 30]
    check(api, oracle_lib, oidx, reads)
    oidx.free()
    # W = 3 and W = 4
    for B in (130, 200):
        gs = [util.random_seq(r, 600) for _ in range(B)]
        oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [(i * 7) % 2 for i in range(B)], ["human", "microbial"],
                                       bin_size=1511, fill_seed=B, fill=0.08)
        check(api, oracle_lib, oidx, util.sample_reads(r, gs, 80, (50, 590), sub_rate=0.02))
        oidx.free()


def test_paired_mode_call_category(api, oracle_lib):
    """paired dehost: mates minimised separately into one entry, call_category (src/dehost_main.cpp:458-470)"""
    r = util.rng(3)
    gs = [util.random_seq(r, 3000) for _ in range(8)]
    cats = ["c%d" % i for i in range(7)] + ["host"]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], list(range(8)), cats)
    m1 = util.sample_reads(r, gs, 400, 150, sub_rate=0.01)
    m2 = [util.mutate(r, gs[int(r.integers(0, 8))][500:650], 0.01) for _ in range(400)]
    m1[0], m2[0] = b"ACGT", b"A" * 150
    m1[1], m2[1] = b"", b"C" * 19
    gpu, _ = check(api, oracle_lib, oidx, m1, m2)
    assert (gpu["call"] != 255).sum() > 50
    # single-end through the fused 8-category path as well (call_host uses categories 0/1 only -> use paired model here)
    oidx.free()
    g2 = [util.random_seq(r, 3000) for _ in range(4)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in g2], [0, 0, 1, 1], ["host", "microbial"])
    m1 = util.sample_reads(r, g2, 150, (20, 200))
    m2 = util.sample_reads(r, g2, 150, (20, 200))
    check(api, oracle_lib, oidx, m1, m2)
    oidx.free()


def test_dense_rows_escape_path(api, oracle_lib):
    """minimisers present in more than three bins (near-identical genomes in 6 bins, heavy background fill): the compact
    row-log entry cannot hold them and the full-row side buffer is used"""
    r = util.rng(71)
    base = util.random_seq(r, 4000)
    gs = [util.mutate(r, base, 0.002 * i) for i in range(6)] + [util.random_seq(r, 4000) for _ in range(4)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1, 0, 1, 0, 1, 0, 1, 0, 1], ["human", "microbial"], bin_size=7001,
                                   fill_seed=9, fill=0.35)
    reads = util.sample_reads(r, gs, 300, (100, 2000), sub_rate=0.01)
    check(api, oracle_lib, oidx, reads)  # the six near-identical genomes put >= 4 set bins into most of their rows
    oidx.free()


def test_many_categories_paired(api, oracle_lib):
    """more than 8 categories: the general row-log path and the array-free model+call kernel (call_category is C-way)"""
    r = util.rng(17)
    gs = [util.random_seq(r, 2500) for _ in range(20)]
    cats = ["cat%d" % i for i in range(11)] + ["human"]
    b2c = [i % 12 for i in range(20)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], b2c, cats, fill_seed=2, fill=0.03)
    m1 = util.sample_reads(r, gs, 250, (80, 300), sub_rate=0.01)
    m2 = util.sample_reads(r, gs, 250, (80, 300), sub_rate=0.01)
    gpu, _ = check(api, oracle_lib, oidx, m1, m2)
    assert len(np.unique(gpu["call"])) > 5
    oidx.free()


def test_other_k_w(api, oracle_lib):
    r = util.rng(4)
    for k, w in ((15, 25), (27, 31), (4, 8), (19, 19), (11, 41)):
        gs = [util.random_seq(r, 3000), util.random_seq(r, 3000)]
        oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "x"], k=k, w=w, bin_size=20011)
        reads = util.sample_reads(r, gs, 100, (1, 800), sub_rate=0.02) + [b"A" * 200, b"AC" * 100]
        check(api, oracle_lib, oidx, reads)
        oidx.free()


def test_non_default_thresholds_and_quality_gate(api, oracle_lib):
    r = util.rng(8)
    gs = [util.random_seq(r, 5000), util.random_seq(r, 5000)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["human", "bug"])
    reads = util.sample_reads(r, gs, 200, (100, 400), sub_rate=0.06)
    g = util.gpu_index_from_oracle(api, oidx)
    seqs, offs, _ = util.concat(reads)
    for mq, conf, minlen in ((10.0, 7, 140), (40.0, 200, 140), (40.0, 0, 300), (40.0, 2, 0)):
        thr = oracle_lib.default_thresholds()
        thr.confidence_threshold, thr.min_length = conf, minlen
        orc = oidx.process_reads(seqs, offs, mq_const=mq, thr=thr)
        ct = conf - 256 if conf >= 128 else conf  # int8 narrowing (include/classify_stats.hpp:404,497)
        m = api.default_model(2, 0, confidence_threshold=ct, min_length=minlen)
        gpu = run_gpu(api, g, reads, mq=mq, comp=0.0, model=m)
        util.assert_parity(gpu, orc)
    g.destroy()
    oidx.free()


def test_device_fabricated_workload_matches_oracle(api, oracle_lib):
    """bench.py's synthetic index + reads are made on the device; pull them back and replay them through the oracle"""
    from charon_amd import pack
    B, S = 6, 1 << 16
    b2c = [0, 1, 0, 1, 0, 1]
    desc = api.make_desc(B, S, b2c, 2, 0)
    g = api.Index(desc)
    n_gen, glen = 6, 8192
    gen = api.synth_genomes(0, 43, n_gen, glen)
    g.synth_fill(43, 0.05)
    g.synth_plant(gen, n_gen, glen, list(range(6)))
    words = g.download()
    oidx = oracle_lib.Index.new(B, S, b2c, ["human", "microbial"])
    # independent check of the emplace kernel: plant the same genomes through the oracle on top of the same fill
    genomes = pack.unpack_reads(api.device_download(0, gen, n_gen * glen // 4, np.uint32), [i * glen for i in range(n_gen)],
                                [glen] * n_gen)
    g2 = api.Index(desc)
    g2.synth_fill(43, 0.05)
    oidx.words()[:] = g2.download()
    g2.destroy()
    fill_density = np.unpackbits(oidx.words().view(np.uint8)).mean() * 64 / B
    assert abs(fill_density - 0.05) < 0.005
    for b, gs in enumerate(genomes):
        oidx.emplace_many(np.unique(oracle_lib.minimisers(gs.decode())), b)
    assert np.array_equal(oidx.words(), words)
    rd = api.synth_reads(0, 42, gen, n_gen, glen, 500, 300, 3000, 0.05, 0.1, 40.0, first_read_id=1000)
    lens = api.device_download(0, rd.seg1_length, 500 * 4, np.uint32)
    offs = api.device_download(0, rd.seg1_offset, 500 * 8, np.uint64)
    assert lens.min() >= 300 and lens.max() <= 3000 and (offs % 64 == 0).all()
    reads = pack.unpack_reads(api.device_download(0, rd.bases2, rd.n_bases // 4, np.uint32), offs, lens)
    st = api.Stream(g, 500, rd.n_bases)
    st.set_model(api.default_model(2, 0))
    st.submit_device(500, rd.n_bases, rd.bases2, rd.seg1_offset, rd.seg1_length, rd.mean_quality, rd.compression)
    res = st.wait_device()
    gpu = dict(num_hashes=api.device_download(0, res.num_hashes, 500 * 4, np.uint32),
               counts=api.device_download(0, res.counts, 500 * 2 * 4, np.uint32).reshape(500, 2),
               unique=api.device_download(0, res.unique_counts, 500 * 2 * 4, np.uint32).reshape(500, 2),
               probs=api.device_download(0, res.probabilities, 500 * 2 * 8, np.float64).reshape(500, 2),
               call=api.device_download(0, res.call, 500, np.uint8), conf=api.device_download(0, res.confidence, 500, np.uint8))
    orc = run_oracle(oidx, reads)
    util.assert_parity(gpu, orc)
    assert (gpu["call"] == 0).sum() > 50 and (gpu["call"] == 1).sum() > 50
    st.destroy()
    for p in (rd.bases2, rd.seg1_offset, rd.seg1_length, rd.mean_quality, rd.compression, gen):
        api.device_free(0, p)
    g.destroy()
    oidx.free()


def test_batch_properties_order_and_repeat(api, oracle_lib):
    """size-independent properties: a permuted batch gives permuted results; re-running is idempotent"""
    r = util.rng(12)
    gs = [util.random_seq(r, 20000), util.random_seq(r, 20000)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "microbial"])
    g = util.gpu_index_from_oracle(api, oidx)
    reads = util.sample_reads(r, gs, 3000, (200, 6000), sub_rate=0.05)
    a = run_gpu(api, g, reads, comp=0.0)
    b = run_gpu(api, g, reads, comp=0.0)
    perm = r.permutation(len(reads))
    c = run_gpu(api, g, [reads[i] for i in perm], comp=0.0)
    for key in ("num_hashes", "counts", "unique", "call", "conf", "probs"):
        assert np.array_equal(a[key], b[key], equal_nan=True) if key == "probs" else np.array_equal(a[key], b[key])
        assert np.array_equal(a[key][perm], c[key], equal_nan=True) if key == "probs" else np.array_equal(a[key][perm], c[key])
    # spot-check 300 of them against the oracle
    pick = r.choice(len(reads), 300, replace=False)
    orc = run_oracle(oidx, [reads[i] for i in pick])
    util.assert_parity({k: v[pick] for k, v in a.items() if k != "flags"}, orc)
    g.destroy()
    oidx.free()


def test_row_sharded_mode_equals_replica_mode(api, oracle_lib):
    """north-star config 4 on one GPU: two row shards of the index, per-shard partial probe words, their SUM (what the
    RCCL all-reduce computes; exactly one shard owns each row) fed back -> identical to the whole-index path and the oracle"""
    from charon_amd import pack
    r = util.rng(31)
    for B, cats in ((2, [0, 1]), (70, [i % 2 for i in range(70)])):
        gs = [util.random_seq(r, 2500) for _ in range(B)]
        oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], cats, ["host", "microbial"], bin_size=30011, fill_seed=3, fill=0.05)
        reads = util.sample_reads(r, gs, 300, (50, 2000), sub_rate=0.03) + [b"A" * 300, b""]
        orc = run_oracle(oidx, reads)
        S, Wd = oidx.bin_size, oidx.bin_words
        cut = S // 3
        words = oidx.words()
        shards = []
        for lo, hi in ((0, cut), (cut, S)):
            d = api.make_desc(oidx.bins, S, oidx.bin_to_cat, 2, 0, row_begin=lo, row_end=hi)
            sh = api.Index(d)
            sh.upload(words[lo * Wd:hi * Wd], row_begin=lo)
            shards.append(sh)
        p = pack.pack_reads(reads)
        n = len(reads)
        # one stream per shard, as two ranks would have: both minimise the same batch independently, so the entry order of
        # the two partial buffers must agree (deterministic length ordering + emission order)
        sts = [api.Stream(sh, n, p["n_bases"]) for sh in shards]
        for st in sts:
            st.set_model(api.default_model(2, 0))
        Es = [st.shard_minimise_host(p) for st in sts]
        E = Es[0]
        assert Es[0] == Es[1] == int(orc["num_hashes"].sum())
        nwords = E * 3 * Wd
        bufs = [api.device_malloc(0, nwords * 8) for _ in shards]
        for st, sh, buf in zip(sts, shards, bufs):
            st.shard_probe(sh, buf, nwords)
        parts = [api.device_download(0, buf, nwords * 8, np.uint64) for buf in bufs]
        # a word is non-zero in at most one shard's partial buffer
        assert not np.any((parts[0] != 0) & (parts[1] != 0))
        total = parts[0] + parts[1]
        outs = []
        for st, buf in zip(sts, bufs):
            api.device_upload(0, buf, total)
            st.shard_finish(buf)
            outs.append(st.wait_host())
        gpu = outs[0]
        for key in ("num_hashes", "counts", "unique", "call", "conf"):
            assert np.array_equal(outs[0][key], outs[1][key])
        st = sts[0]
        sts[1].destroy()
        # mean quality / compression were not passed: gate on the same values in the oracle comparison
        thr = oracle_lib.default_thresholds()
        seqs, offs, _ = util.concat(reads)
        orc0 = oidx.process_reads(seqs, offs, mq_const=0.0, thr=thr)
        util.assert_parity(gpu, orc0)
        for buf in bufs:
            api.device_free(0, buf)
        st.destroy()
        for sh in shards:
            sh.destroy()
        oidx.free()


def test_randomised_parameter_sweep(api, oracle_lib):
    """many small random configurations: k, w, bin count / category map (single- and multi-bin, W = 1..3), read lengths,
    N density, low-complexity inserts, paired or single -- every integer column bit-exact against the oracle"""
    r = util.rng(2024)
    for trial in range(24):
        k = int(r.integers(3, 28))
        w = int(k + r.integers(0, 30))
        B = int(r.choice([1, 2, 3, 5, 9, 33, 64, 65, 130]))
        C = int(min(B, r.integers(2, 7))) if B > 1 else 1
        b2c = [int(x) for x in r.integers(0, C, B)]
        for c in range(C):  # every category owns at least one bin
            b2c[c % B] = c
        if B == 1:
            continue  # call_host / call_category need >= 2 categories
        cats = ["human"] + ["c%d" % i for i in range(1, C)]
        gs = [util.random_seq(r, int(r.integers(300, 3000))) for _ in range(B)]
        oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], b2c, cats, k=k, w=w, bin_size=int(r.integers(2000, 60000)),
                                       fill_seed=trial, fill=float(r.choice([0.0, 0.05, 0.2])))
        n = int(r.integers(1, 150))
        paired = bool(r.integers(0, 2)) or C > 2  # single-end call_host uses categories 0/1 only; fine for C == 2
        reads = util.sample_reads(r, gs, n, (0, int(r.integers(5, 1200))), sub_rate=0.03, random_fraction=0.2)
        mates = util.sample_reads(r, gs, n, (0, int(r.integers(5, 400))), sub_rate=0.03) if paired else None

        def spice(s):
            a = bytearray(s)
            if len(a) > 10 and r.random() < 0.3:
                for p in r.integers(0, len(a), int(r.integers(1, 6))):
                    a[int(p)] = ord("N")
            if len(a) > 60 and r.random() < 0.2:
                p = int(r.integers(0, len(a) - 50))
                a[p:p + 50] = (b"AT" * 25) if r.random() < 0.5 else b"G" * 50
            return bytes(a)
        reads = [spice(s) for s in reads]
        if mates:
            mates = [spice(s) for s in mates]
        check(api, oracle_lib, oidx, reads, mates)
        oidx.free()


def test_long_reads(api, oracle_lib):
    """reads far longer than a wavefront's usual share (250 kb next to 100 b in the same batch)"""
    r = util.rng(99)
    gs = [util.random_seq(r, 300000), util.random_seq(r, 300000)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "microbial"])
    reads = [util.mutate(r, gs[0][1000:251000], 0.05), gs[1][:100], util.mutate(r, gs[1][5:120005], 0.1), b"ACGT" * 20000] + \
        util.sample_reads(r, gs, 70, (100, 3000))
    gpu, orc = check(api, oracle_lib, oidx, reads)
    assert gpu["num_hashes"][0] > 20000 and gpu["conf"][0] == 255
    oidx.free()


def test_full_size_batch_split_invariance(api, oracle_lib):
    """BASELINE-sized inputs, size-independent property: 262 144 device-fabricated 5 kb reads against a 1 GiB 100-bin index
    classified as ONE batch must equal the same reads classified as four batches (different wavefront grouping, different
    row-log layout), and a 512-read sample must equal the oracle."""
    from charon_amd import pack
    B, S, n, L = 100, 1 << 26, 1 << 18, 5000
    b2c = [b % 2 for b in range(B)]
    g = api.Index(api.make_desc(B, S, b2c, 2, 0))
    gen = api.synth_genomes(0, 43, B, 1 << 18)
    g.synth_fill(43, 0.215)
    g.synth_plant(gen, B, 1 << 18, list(range(B)))

    def run(first, count):
        rd = api.synth_reads(0, 42, gen, B, 1 << 18, count, L, L, 0.05, 0.1, 40.0, first_read_id=first)
        st = api.Stream(g, count, rd.n_bases)
        st.set_model(api.default_model(2, 0))
        st.submit_device(count, rd.n_bases, rd.bases2, rd.seg1_offset, rd.seg1_length, rd.mean_quality, rd.compression)
        res = st.wait_device()
        out = dict(num_hashes=api.device_download(0, res.num_hashes, count * 4, np.uint32),
                   counts=api.device_download(0, res.counts, count * 8, np.uint32).reshape(count, 2),
                   unique=api.device_download(0, res.unique_counts, count * 8, np.uint32).reshape(count, 2),
                   probs=api.device_download(0, res.probabilities, count * 16, np.float64).reshape(count, 2),
                   call=api.device_download(0, res.call, count, np.uint8), conf=api.device_download(0, res.confidence, count, np.uint8))
        sample = None
        if first == 0:
            lens = api.device_download(0, rd.seg1_length, 512 * 4, np.uint32)
            offs = api.device_download(0, rd.seg1_offset, 512 * 8, np.uint64)
            nb = int(offs[-1]) + 5056
            sample = pack.unpack_reads(api.device_download(0, rd.bases2, nb // 4, np.uint32), offs, lens)
        st.destroy()
        for p in (rd.bases2, rd.seg1_offset, rd.seg1_length, rd.mean_quality, rd.compression):
            api.device_free(0, p)
        return out, sample

    whole, sample = run(0, n)
    parts = [run(i * (n // 4), n // 4)[0] for i in range(4)]
    for key in whole:
        cat = np.concatenate([p[key] for p in parts])
        assert np.array_equal(whole[key], cat, equal_nan=(key == "probs")) if key == "probs" else np.array_equal(whole[key], cat), key
    assert (whole["call"] == 0).sum() > n // 3 and (whole["call"] == 1).sum() > n // 3
    # oracle on a sample (index words pulled back from the device)
    oidx = oracle_lib.Index.new(B, S, b2c, ["human", "microbial"])
    oidx.words()[:] = g.download()
    orc = run_oracle(oidx, sample)
    util.assert_parity({k: v[:512] for k, v in whole.items()}, orc)
    oidx.free()
    api.device_free(0, gen)
    g.destroy()


def test_hash_function_counts_and_long_windows(api, oracle_lib):
    """h = 1, 2, 4, 5 hash functions (seqan3 allows 1..5) and windows long enough to need the > 64 KB dynamic-LDS path"""
    r = util.rng(123)
    for nhash in (1, 2, 4, 5):
        gs = [util.random_seq(r, 3000) for _ in range(3)]
        mins = [np.unique(oracle_lib.minimisers(g.decode())) for g in gs]
        oidx = oracle_lib.Index.new(3, 30011, [0, 1, 0], ["host", "bug"], nhash=nhash)
        for b, m in enumerate(mins):
            oidx.emplace_many(m, b)
        assert oidx.hash_funs == nhash
        check(api, oracle_lib, oidx, util.sample_reads(r, gs, 150, (50, 1500), sub_rate=0.03))
        oidx.free()
    for k, w in ((20, 140), (27, 255), (5, 200)):
        gs = [util.random_seq(r, 4000), util.random_seq(r, 4000)]
        oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "bug"], k=k, w=w, bin_size=10007)
        reads = util.sample_reads(r, gs, 130, (1, 3000), sub_rate=0.02) + [b"A" * 700, b"ACGT" * 200, gs[0][:w], gs[0][:w - 1], gs[1][:k]]
        check(api, oracle_lib, oidx, reads)
        oidx.free()


@pytest.mark.parametrize("bins,bin_size,ones,slice_words", [(100, 5000, 60000, 64), (8, 70001, 300000, 1000), (130, 997, 1, 1 << 20),
                                                            (3, 4096, 200000, 7), (200, 3001, 150000, 33)])
def test_device_elias_fano_decode(api, bins, bin_size, ones, slice_words):
    """chn_index_decode_ef (loader): sd_vector low/high arrays -> plain rows on the device, sliced; checked against the positions
    the vector was encoded from (oracle/pyref.ef_encode, the restatement of sdsl's sd_vector layout)."""
    from oracle import pyref
    rng = np.random.default_rng(bins * 7 + ones)
    tb = (bins + 63) // 64 * 64
    universe = tb * bin_size
    # positions only in real bins (bin < bins), strictly increasing
    rows = rng.integers(0, bin_size, ones * 2)
    cols = rng.integers(0, bins, ones * 2)
    pos = np.unique(rows.astype(np.int64) * tb + cols)[:ones]
    wl, lowbits, low, highbits, high = pyref.ef_encode([int(p) for p in pos], universe)
    to_words = lambda v, bits: np.frombuffer(v.to_bytes(((bits + 63) // 64) * 8, "little"), np.uint64).copy()
    g = api.Index(api.make_desc(bins, bin_size, [0] * bins, 1, 0))
    try:
        bad = g.decode_ef(universe, wl, to_words(high, highbits), highbits, to_words(low, lowbits) if wl else np.zeros(0, np.uint64),
                          slice_words=slice_words)
        assert bad == 0
        got = g.download()
        want = np.zeros(universe // 64, np.uint64)
        np.bitwise_or.at(want, pos >> 6, np.uint64(1) << (pos & 63).astype(np.uint64))
        assert np.array_equal(got, want)
        pc = g.bin_popcounts()
        assert np.array_equal(pc, np.bincount(pos % tb, minlength=tb).astype(np.uint64))
    finally:
        g.destroy()
    # a bit in a technical-only bin (>= bins) or beyond m_size is reported, not set
    if bins % 64:
        bad_pos = [int(pos[0] // tb * tb + bins)] if int(pos[0] // tb * tb + bins) not in set(pos.tolist()) else []
        if bad_pos:
            wl, lowbits, low, highbits, high = pyref.ef_encode(bad_pos, universe)
            g = api.Index(api.make_desc(bins, bin_size, [0] * bins, 1, 0))
            try:
                assert g.decode_ef(universe, wl, to_words(high, highbits), highbits, to_words(low, lowbits) if wl else np.zeros(0, np.uint64)) == 1
                assert not g.download().any()
            finally:
                g.destroy()


def test_two_streams_driven_from_two_host_threads(api, oracle_lib):
    """one stream object is not thread-safe, but different stream objects on the same index may be driven from
    different host threads (ctypes releases the GIL during the calls)"""
    import threading
    from charon_amd import pack
    r = util.rng(77)
    gs = [util.random_seq(r, 20000) for _ in range(3)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1, 1], ["host", "microbial"])
    g = util.gpu_index_from_oracle(api, oidx)
    batches = [util.sample_reads(r, gs, 400 + 37 * t, (50, 3000), sub_rate=0.04) for t in range(2)]
    want = [run_oracle(oidx, b) for b in batches]
    errors = []

    def worker(t):
        try:
            reads = batches[t]
            p = pack.pack_reads(reads)
            n = len(reads)
            st = api.Stream(g, n, p["n_bases"])
            st.set_model(api.default_model(2, 0))
            for _ in range(6):
                st.submit_host(p, np.full(n, 40.0, np.float32), np.zeros(n, np.float32))
                st.submit_host(p, np.full(n, 40.0, np.float32), np.zeros(n, np.float32))
                util.assert_parity(st.wait_host(), want[t])
                util.assert_parity(st.wait_host(), want[t])
            st.destroy()
        except BaseException as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    g.destroy()
    oidx.free()
    assert not errors, errors


def test_row_log_overflow_reruns_on_worst_case_buffers(api, oracle_lib):
    """the row log is sized for twice the minimiser density of random sequence; a batch that overruns it is detected on the
    device and re-run by chn_batch_wait on worst-case buffers.  CHN_STREAM_TINY_LOG forces that path; results must not change."""
    from charon_amd import pack
    r = util.rng(55)
    gs = [util.random_seq(r, 3000) for _ in range(70)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [i % 2 for i in range(70)], ["human", "microbial"], bin_size=9001,
                                   fill_seed=8, fill=0.10)  # W = 2; about 1 % of the rows escape (more than three set bins)
    reads = util.sample_reads(r, gs, 500, (50, 2500), sub_rate=0.03) + [b"A" * 900, b"", b"ACGT" * 200]
    orc = run_oracle(oidx, reads)
    g = util.gpu_index_from_oracle(api, oidx)
    p = pack.pack_reads(reads)
    n = len(reads)
    for tiny, two in ((True, False), (True, True), (False, True)):
        st = api.Stream(g, n, p["n_bases"], tiny_log=tiny)
        st.set_model(api.default_model(2, 0))
        mq, cp = np.full(n, 40.0, np.float32), np.zeros(n, np.float32)
        st.submit_host(p, mq, cp)
        if two:
            st.submit_host(p, mq, cp)
        outs = [st.wait_host() for _ in range(2 if two else 1)]
        for o in outs:
            util.assert_parity(o, orc)
        assert st.profile(4)[1] == (len(outs) if tiny else 0)
        st.destroy()
    g.destroy()
    oidx.free()


def test_device_batches_bad_and_overlapping_segments(api, oracle_lib):
    """device batches are not validated by the host: a segment reaching beyond n_bases (or misaligned) must never be read and
    makes chn_batch_wait fail; overlapping segments (chunked references) whose lengths add up to more than n_bases are fine."""
    from charon_amd import pack
    r = util.rng(56)
    gs = [util.random_seq(r, 6000) for _ in range(3)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1, 1], ["host", "microbial"], bin_size=20011)
    g = util.gpu_index_from_oracle(api, oidx)
    p = pack.pack_reads([gs[0]])  # one 6 000-base sequence; segments below are windows into it
    nb = p["n_bases"]
    d_bases = api.device_malloc(0, nb // 4)
    api.device_upload(0, d_bases, p["bases2"])
    offs = np.arange(0, 5760, 64, dtype=np.uint64)          # 90 windows, 64 bases apart ...
    lens = np.minimum(2000, 6000 - offs).astype(np.uint32)  # ... of up to 2 000 bases: sum of lengths >> n_bases
    assert int(lens.sum()) > 10 * nb
    d_off, d_len = api.device_malloc(0, offs.nbytes), api.device_malloc(0, lens.nbytes)
    api.device_upload(0, d_off, offs)
    api.device_upload(0, d_len, lens)
    n = len(offs)
    st = api.Stream(g, n, nb)
    st.set_model(api.default_model(2, 0))
    st.submit_device(n, nb, d_bases, d_off, d_len)
    gpu = util.download_results(api, st.wait_device(), n, 2)
    windows = [gs[0][int(o):int(o) + int(l)] for o, l in zip(offs, lens)]
    seqs, so, _ = util.concat(windows)
    orc = oidx.process_reads(seqs, so, mq_const=0.0)
    util.assert_parity(gpu, orc)
    # now break two segments: one reaches beyond the buffer, one is misaligned
    for bad_off, bad_len in ((offs[5], nb), (offs[7] + 3, 100)):
        o2, l2 = offs.copy(), lens.copy()
        o2[5 if bad_len == nb else 7] = bad_off
        l2[5 if bad_len == nb else 7] = bad_len
        api.device_upload(0, d_off, o2)
        api.device_upload(0, d_len, l2)
        st.submit_device(n, nb, d_bases, d_off, d_len)
        with pytest.raises(api.ChnError, match="misaligned or lies outside"):
            st.wait_device()
        # the deflate pass checks its segments the same way (it must never read them)
        st.submit_device(n, nb, d_bases, d_off, d_len, gzip_tallies=6000, gzip_output=1)
        with pytest.raises(api.ChnError, match="misaligned or lies outside"):
            st.wait_device()
    # the stream stays usable
    api.device_upload(0, d_off, offs)
    api.device_upload(0, d_len, lens)
    st.submit_device(n, nb, d_bases, d_off, d_len)
    util.assert_parity(util.download_results(api, st.wait_device(), n, 2), orc)
    st.destroy()
    for ptr in (d_bases, d_off, d_len):
        api.device_free(0, ptr)
    g.destroy()
    oidx.free()


def test_borderline_reads_are_flagged_and_reevaluated_on_the_host(api, oracle_lib):
    """reads whose deciding probability comparison is a tie (both categories equally likely) carry flags != 0 and are
    re-evaluated by chn_batch_wait with the host libm (host_model_call); the result must equal the oracle's"""
    r = util.rng(57)
    gs = [util.random_seq(r, 5000) for _ in range(4)]
    # single-end: confidence threshold 0 opens the gate for reads without any unique hit (hp == op exactly)
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs[:2]], [0, 1], ["host", "microbial"])
    reads = util.sample_reads(r, gs[:2], 150, (150, 900), sub_rate=0.05, random_fraction=0.4)
    g = util.gpu_index_from_oracle(api, oidx)
    thr = oracle_lib.default_thresholds()
    thr.confidence_threshold = 0
    seqs, offs, _ = util.concat(reads)
    orc = oidx.process_reads(seqs, offs, thr=thr)
    gpu = run_gpu(api, g, reads, comp=0.0, model=api.default_model(2, 0, confidence_threshold=0))
    assert gpu["flags"].sum() > 10
    util.assert_parity(gpu, orc)
    g.destroy()
    oidx.free()
    # paired (call_category): --confidence 255 narrows to -1 (int8), so `conf > threshold` always holds and pf > ps decides
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1, 2, 3], ["host", "a", "b", "c"])
    m1 = util.sample_reads(r, gs, 150, 150, sub_rate=0.02, random_fraction=0.4)
    m2 = util.sample_reads(r, gs, 150, 150, sub_rate=0.02, random_fraction=0.4)
    g = util.gpu_index_from_oracle(api, oidx)
    thr = oracle_lib.default_thresholds(paired=True)
    thr.confidence_threshold = 255
    seqs, offs, split = util.concat(m1, m2)
    orc = oidx.process_reads(seqs, offs, mate_split=split, thr=thr)
    gpu = run_gpu(api, g, m1, m2, comp=0.0, model=api.default_model(4, 0, paired=True, confidence_threshold=-1))
    assert gpu["flags"].sum() > 10
    util.assert_parity(gpu, orc)
    g.destroy()
    oidx.free()


def test_sparse_row_sharded_exchange_equals_replica_mode(api, oracle_lib):
    """north-star config 4, sparse form, rehearsed on one GPU: the index rows are split over three shards (uneven, one of them
    tiny), two 'ranks' each classify THEIR OWN half of the reads: minimise -> per-owner query groups -> (host-side stand-in for
    the all-to-all) -> every owner serves its rows -> rows travel back in query order -> AND / count / call.  Must equal the
    oracle (and hence the whole-index path) bit for bit, for the fused-shaped (W = 1) and the general (W = 2) index."""
    from charon_amd import pack
    r = util.rng(41)
    for B, cats in ((2, [0, 1]), (70, [i % 2 for i in range(70)])):
        gs = [util.random_seq(r, 2500) for _ in range(B)]
        oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], cats, ["host", "microbial"], bin_size=30011, fill_seed=3, fill=0.05)
        reads = util.sample_reads(r, gs, 400, (50, 2000), sub_rate=0.03) + [b"A" * 300, b"", b"ACGT" * 40]
        S, Wd = oidx.bin_size, oidx.bin_words
        splits = [0, 17, S // 3, S]           # three owners
        words = oidx.words()
        shards = []
        for lo, hi in zip(splits[:-1], splits[1:]):
            sh = api.Index(api.make_desc(oidx.bins, S, oidx.bin_to_cat, 2, 0, row_begin=lo, row_end=hi))
            sh.upload(words[lo * Wd:hi * Wd], row_begin=lo)
            shards.append(sh)
        n_own = len(shards)
        halves = [reads[:len(reads) // 2], reads[len(reads) // 2:]]
        ranks = []
        for k, mine in enumerate(halves):     # a "rank" = a stream (on any shard object of the IBF) + its own reads
            p = pack.pack_reads(mine)
            st = api.Stream(shards[k], len(mine), p["n_bases"])
            st.set_model(api.default_model(2, 0))
            st.shardx_minimise_host(p, np.full(len(mine), 40.0, np.float32), np.zeros(len(mine), np.float32))
            n_probes, counts = st.shardx_counts(splits)
            assert n_probes == sum(counts)
            dq = api.device_malloc(0, max(n_probes, 1) * 4)
            st.shardx_queries(dq, n_probes)
            st.sync()
            q = api.device_download(0, dq, n_probes * 4, np.uint32)
            ranks.append(dict(st=st, n=len(mine), n_probes=n_probes, counts=counts, q=q, dq=dq, mine=mine))
        # the "all-to-all": owner o receives, from every rank in rank order, that rank's group o
        back = [np.zeros((rk["n_probes"], Wd), np.uint64) for rk in ranks]
        for o, sh in enumerate(shards):
            groups = [rk["q"][sum(rk["counts"][:o]):sum(rk["counts"][:o + 1])] for rk in ranks]
            qin = np.concatenate(groups)
            assert (qin < splits[o + 1] - splits[o]).all()
            d_in, d_out = api.device_malloc(0, max(qin.size, 1) * 4), api.device_malloc(0, max(qin.size, 1) * Wd * 8)
            api.device_upload(0, d_in, qin)
            ranks[0]["st"].shardx_serve(sh, d_in, qin.size, d_out)   # any stream on the owner's device serves
            ranks[0]["st"].sync()
            rows = api.device_download(0, d_out, qin.size * Wd * 8, np.uint64).reshape(-1, Wd)
            assert np.array_equal(rows, words.reshape(-1, Wd)[splits[o] + qin.astype(np.int64)])
            at = 0
            for rk, grp, bk in zip(ranks, groups, back):   # ... and sends every rank's rows back, in the order they came
                lo = sum(rk["counts"][:o])
                bk[lo:lo + grp.size] = rows[at:at + grp.size]
                at += grp.size
            api.device_free(0, d_in)
            api.device_free(0, d_out)
        seen = 0
        for rk, bk in zip(ranks, back):
            d_back = api.device_malloc(0, max(bk.size, 1) * 8)
            api.device_upload(0, d_back, bk)
            rk["st"].shardx_finish(d_back)
            gpu = rk["st"].wait_host()
            seqs, offs, _ = util.concat(rk["mine"])
            orc = oidx.process_reads(seqs, offs)
            util.assert_parity(gpu, orc)
            assert rk["n_probes"] == int(orc["num_hashes"].sum()) * 3
            seen += rk["n"]
            api.device_free(0, d_back)
            api.device_free(0, rk["dq"])
            rk["st"].destroy()
        assert seen == len(reads)
        for sh in shards:
            sh.destroy()
        oidx.free()


def test_sparse_exchange_with_mismatched_splits_is_an_error(api, oracle_lib):
    """a query that names a row its owner does not hold (queries built from other row_splits than the shards', or delivered to
    the wrong owner) is answered with row 0 by k_shx_serve; the batch in flight on the serving stream must then fail, not return counts."""
    from charon_amd import pack
    r = util.rng(43)
    gs = [util.random_seq(r, 2500) for _ in range(2)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "microbial"], bin_size=30011)
    reads = util.sample_reads(r, gs, 200, (200, 1500))
    S, Wd = oidx.bin_size, oidx.bin_words
    words = oidx.words()
    shards = []
    for lo, hi in ((0, S // 3), (S // 3, S)):
        sh = api.Index(api.make_desc(oidx.bins, S, oidx.bin_to_cat, 2, 0, row_begin=lo, row_end=hi))
        sh.upload(words[lo * Wd:hi * Wd], row_begin=lo)
        shards.append(sh)
    p = pack.pack_reads(reads)
    st = api.Stream(shards[0], len(reads), p["n_bases"])
    st.set_model(api.default_model(2, 0))
    st.shardx_minimise_host(p, np.full(len(reads), 40.0, np.float32), np.zeros(len(reads), np.float32))
    wrong = [0, S // 2, S]                      # not the shards' splits
    n_probes, counts = st.shardx_counts(wrong)
    dq = api.device_malloc(0, max(n_probes, 1) * 4)
    st.shardx_queries(dq, n_probes)
    st.sync()
    q = api.device_download(0, dq, n_probes * 4, np.uint32)
    assert (q[:counts[0]] >= S // 3).any()      # group 0 names rows beyond shard 0
    d_out = api.device_malloc(0, max(n_probes, 1) * Wd * 8)
    st.shardx_serve(shards[0], dq, counts[0], d_out)
    st.shardx_serve(shards[1], dq + 4 * counts[0], counts[1], d_out + 8 * Wd * counts[0])
    st.shardx_finish(d_out)
    with pytest.raises(RuntimeError, match="rows this shard does not hold"):
        st.wait_host()
    api.device_free(0, dq)
    api.device_free(0, d_out)
    st.destroy()
    for sh in shards:
        sh.destroy()
    oidx.free()


def test_gamma_and_beta_models_call_category(api, oracle_lib):
    """`charon classify` / `--dist gamma|beta`: the parametric densities (include/classify_stats.hpp:377-381) in k_model_call, with
    the classify thresholds (include/classify_arguments.hpp:19-29) and call_category, against the oracle; also a moved neg
    location (what GammaParams::fit_loc does) and fitted-looking parameters per category"""
    r = util.rng(61)
    gs = [util.random_seq(r, 5000) for _ in range(3)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1, 2], ["host", "a", "b"])
    reads = util.sample_reads(r, gs, 300, (100, 1500), sub_rate=0.04, random_fraction=0.2)
    g = util.gpu_index_from_oracle(api, oidx)
    seqs, offs, _ = util.concat(reads)
    for dist in ("gamma", "beta"):
        thr = oracle_lib.classify_thresholds(dist=dist)
        thr.min_compression = 0.0  # the gzip column is a host-side input: both sides gate on 0
        orc = oidx.process_reads(seqs, offs, thr=thr)
        m = api.default_model(3, 0, paired=True, dist=dist, min_quality=10.0, min_length=140, min_compression=0.0, confidence_threshold=2,
                              min_proportion_difference=0.0)
        gpu = run_gpu(api, g, reads, comp=0.0, model=m)
        util.assert_parity(gpu, orc)  # host result buffers: the float evaluation of the reference
        assert len(np.unique(gpu["call"])) >= 3 and np.isfinite(gpu["probs"]).all()
        # device-resident results: k_model_call's own evaluation (log-density in double) agrees to ~1e-5 in the probability column
        from charon_amd import pack
        p = pack.pack_reads(reads)
        st = api.Stream(g, len(reads), p["n_bases"])
        st.set_model(m)
        st.submit_host(p, np.full(len(reads), 40.0, np.float32), np.zeros(len(reads), np.float32))
        dev = util.download_results(api, st.wait_device(), len(reads), 3)
        st.destroy()
        util.assert_parity(dev, orc, prob_tol=2e-5)
    # the same through dehost's single-end caller (call_host) on a two-category index: `charon dehost --dist gamma`
    oidx2 = util.build_oracle_index(oracle_lib, [[gs[0]], [gs[1]]], [0, 1], ["host", "microbial"])
    g2 = util.gpu_index_from_oracle(api, oidx2)
    for dist in ("gamma", "beta"):
        thr = oracle_lib.default_thresholds()
        thr.dist = {"gamma": 1, "beta": 2}[dist]
        orc = oidx2.process_reads(seqs, offs, thr=thr)
        gpu = run_gpu(api, g2, reads, comp=0.0, model=api.default_model(2, 0, dist=dist))
        util.assert_parity(gpu, orc)
    g.destroy(); g2.destroy()
    oidx.free(); oidx2.free()


def test_gzip_tallies_through_the_abi(api, oracle_lib):
    """chn_batch.gzip_tallies: per read the literal/length and distance code frequencies of zlib's level-6 deflate block (the CLI
    turns them into the exact gzip size; tests/test_gpu_cli.py compares that column with zlib itself on hundreds of shapes).
    Here: the ABI plumbing and the invariants any correct tally has."""
    from charon_amd import pack
    r = util.rng(66)
    gs = [util.random_seq(r, 4000), util.random_seq(r, 4000)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "microbial"])
    g = util.gpu_index_from_oracle(api, oidx)
    reads = [b"A", b"ACGTN", b"A" * 1000, b"ACGT" * 500, util.random_seq(r, 3000), util.random_seq(r, 9000), util.random_seq(r, 17000), gs[0][:700] * 3]
    p = pack.pack_reads(reads)
    st = api.Stream(g, len(reads), p["n_bases"])
    st.set_model(api.default_model(2, 0, min_compression=0.9))  # a gate that every read would fail ...
    st.submit_host(p, np.full(len(reads), 40.0, np.float32), None, gzip_tallies=16384)
    out = st.wait_host()
    t = out["gzip_tallies"].astype(np.int64)
    lit, lens, dist, status = t[:, :256], t[:, 257:286], t[:, 286:316], t[:, 316]
    assert [int(x != 0) for x in status] == [0, 0, 0, 0, 0, 0, 1, 0]  # 17 000 letters: beyond the device's bound -> sized on the host
    ok = status == 0
    assert (lens.sum(1) == dist.sum(1))[ok].all()               # every match has one length and one distance code
    assert (lit[:, [65, 67, 71, 84, 78]].sum(1) == lit.sum(1))[ok].all()  # only A C G T N literals
    assert lit[0, 65] == 1 and lens[0].sum() == 0
    assert lit[1].sum() == 5 and lens[1].sum() == 0
    assert lit[2, 65] >= 1 and lens[2, 28] >= 3                  # a homopolymer: one literal, then maximal (258) matches at distance 1
    assert dist[2, 0] == lens[2].sum()
    for i in (4, 5):                                            # random four-letter sequence: nearly everything is a short match
        n = len(reads[i])
        assert 0 < lit[i].sum() < 0.3 * n and lens[i, :10].sum() >= 0.9 * lens[i].sum() and lit[i].sum() + 3 * lens[i].sum() <= n
    assert lit[7].sum() < 900 and lens[7, 28] >= 4              # a threefold repeat: the 2nd and 3rd copy are long matches
    # ... is left open by the call kernel when the ratios are still to come (the caller applies it): calls survive
    assert out["call"][7] == 0  # the repeat of the host genome is called although 0 < min_compression 0.9
    st.destroy()
    g.destroy()
    oidx.free()


def test_three_batches_in_flight_come_back_in_order(api, oracle_lib):
    """Up to three batches may be in flight (a caller with host buffers keeps the next upload under the running probe kernel);
    chn_batch_wait hands them back oldest first, each equal to the same batch run alone; a fourth submit is refused."""
    from charon_amd import pack
    r = util.rng(91)
    gs = [util.random_seq(r, 30000), util.random_seq(r, 30000)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "microbial"])
    g = util.gpu_index_from_oracle(api, oidx)
    sets = [util.sample_reads(r, gs, n, (150, 3000), sub_rate=0.04) for n in (700, 1100, 400, 900, 650)]
    alone = [run_gpu(api, g, s, comp=0.0) for s in sets]
    packed = [pack.pack_reads(s) for s in sets]
    cap = max(len(s) for s in sets)
    st = api.Stream(g, cap, max(p["n_bases"] for p in packed))
    st.set_model(api.default_model(2, 0))
    sub = lambda i: st.submit_host(packed[i], np.full(len(sets[i]), 40.0, np.float32), np.zeros(len(sets[i]), np.float32))
    sub(0); sub(1); sub(2)
    with pytest.raises(RuntimeError, match="in flight"):
        sub(3)
    got = [st.wait_host()]
    sub(3)
    got.append(st.wait_host())
    sub(4)
    got += [st.wait_host(), st.wait_host(), st.wait_host()]
    for a, b in zip(alone, got):
        for key in ("num_hashes", "counts", "unique", "call", "conf"):
            assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a["probs"], b["probs"], equal_nan=True)
    st.destroy()
    g.destroy()
    oidx.free()


def test_gzip_sizes_on_the_device_equal_zlib(api, oracle_lib):
    """chn_batch.gzip_output = sizes: deflate pass AND _tr_flush_block's tree arithmetic on the device; the number that comes back
    is the byte count of the gzip member zlib writes for the read's letters (level 6, as gzip-hpp does: src/utils.cpp:114-124),
    for every shape of read -- stored, static and dynamic blocks, forced second code, bit-length overflow"""
    import zlib
    from charon_amd import pack
    r = util.rng(67)
    gs = [util.random_seq(r, 4000), util.random_seq(r, 4000)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "microbial"])
    g = util.gpu_index_from_oracle(api, oidx)
    reads = [b"A", b"AC", b"ACG", b"ACGTN", b"A" * 1000, b"ACGT" * 500, b"N" * 300, b"AAC" * 700, gs[0][:700] * 3, util.random_seq(r, 17000), util.random_seq(r, 62000),
             b"A" * 50000, b"ACGTTGCA" * 7000]
    for n in (4, 7, 10, 16, 33, 70, 150, 400, 1000, 2500, 5000, 9000, 16384, 24000, 40000, 61440):
        reads.append(util.random_seq(r, n))
        reads.append(bytes(r.choice(list(b"ACGTN"), n, p=[0.3, 0.2, 0.2, 0.2, 0.1]).astype(np.uint8)))
    for k in range(60):  # low-entropy and skewed compositions: long codes, few symbols
        n = int(r.integers(20, 6000))
        p = r.dirichlet([0.3] * 4)
        unit = bytes(r.choice(list(b"ACGT"), int(r.integers(1, 40)), p=p).astype(np.uint8))
        s = bytearray((unit * (n // len(unit) + 1))[:n])
        for _ in range(int(r.integers(0, 30))):
            s[int(r.integers(0, n))] = b"ACGT"[int(r.integers(0, 4))]
        reads.append(bytes(s))

    def zsize(b):
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 8)
        return len(co.compress(b) + co.flush())
    # a batch without a single N takes the kernel's other form (codes of two bits, sixteen letters per comparison, no end mark in LDS)
    for reads in (reads, [rd.replace(b"N", b"G") for rd in reads]):
        p = pack.pack_reads(reads)
        st = api.Stream(g, len(reads), p["n_bases"])
        st.set_model(api.default_model(2, 0))
        st.submit_host(p, np.full(len(reads), 40.0, np.float32), None, gzip_tallies=61440, gzip_output=2)
        out = st.wait_host()
        sizes, status = out["gzip_sizes"], out["gzip_tallies"][:, 316]
        on_device = 0
        for i, rd in enumerate(reads):
            if len(rd) > 61440:
                assert status[i] != 0 and sizes[i] == 0
                continue
            if status[i] != 0:  # more than 16 382 symbols: a second deflate block -- only long reads with few matches may say so
                assert len(rd) > 16383 * 3 // 2 and sizes[i] == 0, (i, len(rd))
                continue
            on_device += 1
            assert int(sizes[i]) == zsize(rd), (i, len(rd), rd[:40])
        assert on_device >= len(reads) - 6
        # sizes only: no tallies come back, the same numbers do
        st.submit_host(p, np.full(len(reads), 40.0, np.float32), None, gzip_tallies=61440, gzip_output=1)
        out1 = st.wait_host()
        assert "gzip_tallies" not in out1 and np.array_equal(out1["gzip_sizes"], sizes)
        st.destroy()
    g.destroy()
    oidx.free()


def test_gzip_sizes_when_a_wavefront_takes_many_reads(api, oracle_lib):
    """k_gzip_tally's wavefronts take read after read from a counter and re-use their scratch (class arrays in global memory) and LDS:
    a batch of far more reads than the device holds wavefronts (8 192 on 256 CUs), every size against zlib"""
    import zlib
    from charon_amd import pack
    r = util.rng(71)
    gs = [util.random_seq(r, 4000), util.random_seq(r, 4000)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "microbial"])
    g = util.gpu_index_from_oracle(api, oidx)
    n = 40000
    pool = util.random_seq(r, 200000)
    starts, lens = r.integers(0, 199000, n), r.integers(1, 700, n)
    reads = []
    for i in range(n):
        rd = pool[int(starts[i]):int(starts[i]) + int(lens[i])]
        if i % 7 == 0:
            rd = rd[:len(rd) // 2] * 2 + rd[:3]      # an internal copy: long matches
        reads.append(rd)

    def zsize(b):
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 8)
        return len(co.compress(b) + co.flush())
    for with_n in (False, True):  # both forms of the kernel (a batch with an N keeps four-bit codes)
        if with_n:
            reads[5] = reads[5][:10] + b"N" + reads[5][10:]
        p = pack.pack_reads(reads)
        st = api.Stream(g, n, p["n_bases"])
        st.set_model(api.default_model(2, 0))
        st.submit_host(p, np.full(n, 40.0, np.float32), None, gzip_tallies=61440, gzip_output=1)
        sizes = st.wait_host()["gzip_sizes"]
        st.destroy()
        want = np.array([zsize(rd) for rd in reads], np.uint32)
        bad = np.nonzero(sizes != want)[0]
        assert bad.size == 0, (with_n, bad[:10], sizes[bad[:10]], want[bad[:10]])
    g.destroy()
    oidx.free()


def test_bin_popcounts_of_a_large_index(api):
    """chn_index_bin_popcounts (the loader's last self-check) on an index large enough that every thread of the kernel walks more than
    255 words of its column (its byte-sliced accumulators are emptied on the way): 2^28 + 777 rows of one word, against numpy"""
    S = (1 << 28) + 777
    g = api.Index(api.make_desc(5, S, [0, 1, 0, 1, 0], 2, 0))
    try:
        g.synth_fill(43, 0.215)
        pc = g.bin_popcounts()
        w = g.download()
        want = np.array([int(((w >> np.uint64(b)) & np.uint64(1)).sum()) for b in range(5)], np.uint64)
        assert np.array_equal(pc[:5], want) and not pc[5:].any() and not (w >> np.uint64(5)).any()
        assert 0.2 * S < want.min() and want.max() < 0.23 * S
    finally:
        g.destroy()


def test_gather_roof_is_a_plausible_rate(api, oracle_lib):
    """chn_index_gather_roof (what bench.py prices the probe kernel against): random row fetches per second of this device on this
    index -- a finite positive rate, the same order of magnitude for both cache policies (a toy index sits in the caches: far above the
    50 G/s a 39 GB table gives)"""
    r = util.rng(3)
    gs = [util.random_seq(r, 3000), util.random_seq(r, 3000)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "microbial"])
    g = util.gpu_index_from_oracle(api, oidx)
    a, b = g.gather_roof(nt=False), g.gather_roof(nt=True)
    assert 1e9 < a < 1e13 and 1e9 < b < 1e13 and 0.2 < a / b < 5.0
    g.destroy()
    oidx.free()
