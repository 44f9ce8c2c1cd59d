"""CPU tests: the oracle against every known-answer vector available for this path.

The reference ships no tests or fixtures, so the pins are: seqan3's public documentation
vectors, the surveyor's cross-check values (A.8), and bit-for-bit agreement between the two
independent restatements (C++ oracle/charon_oracle.hpp, python oracle/pyref.py).
"""
import os

import numpy as np
import pytest

from oracle import pyref
from tests import util

ROOT = util.ROOT


def test_kmer_hash_doc_vector(oracle_lib):
    assert list(oracle_lib.kmer_hashes("ACGTAGC", 3, sigma=4)) == [6, 27, 44, 50, 9]


def test_minimiser_hash_doc_vector(oracle_lib):
    want = [10322096095657499224, 10322096095657499142, 10322096095657499224]
    assert [int(x) for x in oracle_lib.minimisers("CCACGTCGACGGTT", 4, 8, sigma=4)] == want
    assert pyref.minimisers("CCACGTCGACGGTT", 4, 8, sigma=4) == want
    assert want[0] == pyref.SEED ^ 134 and want[1] == pyref.SEED ^ 216


def test_ibf_doc_smoke(oracle_lib):
    idx = oracle_lib.Index.new(12, 8192, [0] * 12, ["x"], nhash=2)
    for v, b in ((126, 0), (712, 3), (237, 9)):
        idx.emplace_many(np.array([v], np.uint64), b)
    row = int(idx.bulk_contains(712)[0])
    assert [(row >> b) & 1 for b in range(12)] == [0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0]


def test_survey_cross_check_my_fasta(oracle_lib, my_fasta):
    want = {"KX789432.1": (859, 74), "KJ943213.1": (2205, 195), "JQ431377.1": (2274, 206), "JQ431390.1": (2280, 210),
            "JF429396.1": (1701, 159), "JQ431260.1": (1410, 123)}
    allm = set()
    for name, (L, M) in want.items():
        m = oracle_lib.minimisers(my_fasta[name])
        assert (len(my_fasta[name]), len(m)) == (L, M)
        assert len(set(int(x) for x in m)) == M
        allm.update(int(x) for x in m)
    assert len(allm) == 967
    assert oracle_lib.lib().orc_bin_size_in_bits(967, 3, 0.01) == 11957 == pyref.bin_size_in_bits(967)
    m = oracle_lib.minimisers(my_fasta["KX789432.1"])
    assert [int(x) for x in m[:3]] == [10322078109994799690, 10322077211472010273, 10322077590812346780]
    assert int(m[-1]) == 10322077187095267847


def test_hash_and_fit_cross_check(oracle_lib):
    v = 10322078109994799690
    assert [oracle_lib.lib().orc_hash_and_fit(v, i, 1 << 27) for i in range(3)] == [114661838, 63461117, 56320885]
    assert [oracle_lib.lib().orc_hash_and_fit(v, i, 12000) for i in range(3)] == [2226, 7692, 899]
    assert [pyref.hash_and_fit(v, i, 12000) for i in range(3)] == [2226, 7692, 899]


def test_gzip_ratio_cross_check(oracle_lib, my_fasta):
    want = [0.379511, 0.336054, 0.340809, 0.335088, 0.343327, 0.347518]
    got = [float(oracle_lib.lib().orc_compression_ratio(s.encode(), len(s))) for s in my_fasta.values()]
    assert np.allclose(got, want, atol=5e-7)
    assert np.allclose([float(pyref.gzip_ratio(s)) for s in my_fasta.values()], got, atol=0)


def test_default_kde_cross_check(oracle_lib):
    want = {0: 9.26653e-05, 0.01: 0.00136087, 0.02: 0.00166912, 0.05: 0.0118848, 0.1: 0.988947, 0.15: 0.999995, 0.2: 1, 0.3: 1,
            0.5: 1, 0.8: 1, 1.0: 1}
    for x, p in want.items():
        got = oracle_lib.lib().orc_default_model_prob(x, 0)
        assert abs(got - p) <= 6e-6 * max(p, 1e-4)
    comps = [oracle_lib.lib().orc_default_model_prob(0.05, i) for i in (4, 2, 3)]
    assert np.allclose(comps, [9.17707e-05, 0.085844, 7.13705], rtol=2e-6)


def test_model_prob_two_restatements_agree(oracle_lib):
    tables = pyref.load_tables(os.path.join(ROOT, "charon_amd", "data", "default_kde.txt"))
    for x in [0.0, 0.003, 0.0125, 0.049, 0.051, 0.0999, 0.25, 0.75, 1.0]:
        pos, neg = pyref.model_prob(np.float32(x), tables)
        assert pos == oracle_lib.lib().orc_default_model_prob(x, 0)
        assert neg == oracle_lib.lib().orc_default_model_prob(x, 1)


EDGE_SEQS = ["A" * 1000, "ACGT" * 40, "AC" * 100, "ACG" * 60 + "T" * 50, "A" * 18, "A" * 19, "ACGTTGCA" * 3 + "GATTACA",
             "ACGTNNNNACGTACGTACGTAGCTAGCTAGCATCGATCGATCAGCTACGATCGATCGACTAGCTAGCTAGCTAGCATGCATCGATGCATGCAT",
             "N" * 100, "acgtRYKMacgtacgtagctagctagcatcgatcgatcagctacgatcgatcgactagctagctagct", ""]


def test_minimisers_two_restatements_agree_on_edges(oracle_lib, my_fasta):
    r = util.rng(7)
    seqs = EDGE_SEQS + [util.random_seq(r, int(n)).decode() for n in (19, 20, 40, 41, 42, 63, 64, 65, 300)] + \
        [my_fasta["KX789432.1"][:400]]
    for s in seqs:
        for k, w in ((19, 41), (4, 8), (15, 15 + 9), (27, 31)):
            a = [int(x) for x in oracle_lib.minimisers(s, k, w)]
            assert a == pyref.minimisers(s, k, w), (s[:30], k, w)
    assert len(oracle_lib.minimisers("A" * 1000)) == 42  # SURVEY A.8 tie path
    assert len(oracle_lib.minimisers("A" * 18)) == 0


def test_counts_two_restatements_agree(oracle_lib):
    r = util.rng(11)
    g = [[util.random_seq(r, 3000)], [util.random_seq(r, 2500)], [util.random_seq(r, 2000)], [util.random_seq(r, 2800)]]
    b2c = [0, 1, 0, 1]
    oidx = util.build_oracle_index(oracle_lib, g, b2c, ["host", "microbial"], bin_size=4099)
    pib = pyref.PlainIBF(4, 4099)
    pib.data[:] = oidx.words()
    reads = util.sample_reads(r, [x[0] for x in g], 12, 400, sub_rate=0.03)
    seqs, offs, _ = util.concat(reads)
    o = oidx.process_reads(seqs, offs)
    for i, rd in enumerate(reads):
        nh, counts, unique, props, uprops = pyref.count_read(pib, pyref.minimisers(rd.decode()), b2c, 2)
        assert nh == o["num_hashes"][i]
        assert counts == list(o["counts"][i]) and unique == list(o["unique"][i])
        assert [float(x) for x in uprops] == [float(x) for x in o["uprops"][i]]
    oidx.free()


def test_sd_vector_matches_plain_and_file_roundtrip(oracle_lib, tmp_path):
    r = util.rng(3)
    g = [[util.random_seq(r, 5000)] for _ in range(70)]  # 70 bins -> two words per row
    oidx = util.build_oracle_index(oracle_lib, g, [i % 2 for i in range(70)], ["human", "microbial"], bin_size=2003)
    plain = oidx.words().copy()
    oidx.compress()
    assert oidx.ef_ones == int(sum(bin(int(x)).count("1") for x in plain))
    for row in list(range(0, 2003, 97)) + [2002]:
        for wd in range(2):
            assert oracle_lib.lib().orc_sd_get_int(oidx.h, row * 128 + 64 * wd) == int(plain[row * 2 + wd])
    # unaligned get_int
    bits = "".join(format(int(x), "064b")[::-1] for x in plain)
    for bit in (1, 63, 65, 127, 1000, 2003 * 128 - 64):
        assert oracle_lib.lib().orc_sd_get_int(oidx.h, bit) == int(bits[bit:bit + 64][::-1], 2)
    path = str(tmp_path / "t.idx")
    oidx.store(path)
    back = oracle_lib.Index.load(path)
    assert np.array_equal(back.words(), plain)
    assert (back.k, back.w, back.bins, back.bin_size, back.bin_words, back.ncat) == (19, 41, 70, 2003, 2, 2)
    assert back.categories == ["human", "microbial"] and list(back.bin_to_cat) == [i % 2 for i in range(70)]
    # EF probes (the reference's in-RAM form) give the same answers as plain probes
    mins = oracle_lib.minimisers(g[5][0].decode())
    for v in mins[:50]:
        back.use_ef(False)
        a = back.bulk_contains(int(v))
        back.use_ef(True)
        assert np.array_equal(a, back.bulk_contains(int(v)))
    # the python codec reads the C++ file and vice versa
    py = pyref.read_index(path)
    assert np.array_equal(py["ibf"].data, plain) and py["categories"] == ["human", "microbial"] and py["k"] == 19
    path2 = str(tmp_path / "p.idx")
    pyref.write_index(path2, py["k"], py["w"], py["max_fpr"], py["categories"], py["filepath_to_bin"], py["bin_to_category"],
                      py["num_files"], py["records_per_bin"], py["hashes_per_bin"], py["ibf"])
    assert open(path, "rb").read() == open(path2, "rb").read()
    back.free()
    oidx.free()


def test_call_host_two_restatements_agree(oracle_lib):
    r = util.rng(5)
    tables = pyref.load_tables(os.path.join(ROOT, "charon_amd", "data", "default_kde.txt"))
    g = [[util.random_seq(r, 6000)], [util.random_seq(r, 6000)]]
    oidx = util.build_oracle_index(oracle_lib, g, [0, 1], ["microbial", "host"])
    reads = util.sample_reads(r, [x[0] for x in g], 40, (150, 900), sub_rate=0.08, random_fraction=0.2)
    seqs, offs, _ = util.concat(reads)
    o = oidx.process_reads(seqs, offs, mq_const=30.0)
    assert oidx.host_index == 1
    seen = set()
    for i in range(len(reads)):
        probs = [pyref.model_prob(o["uprops"][i][c], tables)[0] for c in range(2)]
        assert probs == list(o["probs"][i])
        call, conf = pyref.call_host(o["unique"][i].tolist(), o["uprops"][i], probs, 1, 30.0, len(reads[i]), 0.0)
        assert (call, conf) == (int(o["call"][i]), int(o["conf"][i]))
        seen.add(call)
    assert seen >= {0, 1}
    oidx.free()


def test_first_read_dropped_without_extract_and_tsv_shape(oracle_lib, tmp_path):
    """quirk list A.9: without --extract the first read is neither cached nor printed (include/result.hpp:80-85,139-151)"""
    r = util.rng(9)
    g = [[util.random_seq(r, 4000)], [util.random_seq(r, 4000)]]
    oidx = util.build_oracle_index(oracle_lib, g, [0, 1], ["microbial", "host"])
    reads = util.sample_reads(r, [x[0] for x in g], 7, 300)
    fq = tmp_path / "r.fastq"
    with open(fq, "w") as f:
        for i, s in enumerate(reads):
            f.write("@r%d extra words\n%s\n+\n%s\n" % (i, s.decode(), "I" * len(s)))
    tsv = oidx.dehost_files(str(fq))
    rows = tsv.strip().split("\n")
    assert [x.split("\t")[1] for x in rows] == ["r%d" % i for i in range(1, 7)]
    f0 = rows[0].split("\t")
    assert f0[0] in "CU" and f0[3] == "300" and f0[5] == "40" and len(f0) == 9 and f0[8].endswith(" ")
    assert f0[8].split(" ")[0].count(":") == 4
    # with --extract the cache has capacity and every read is printed at complete()
    rows2 = oidx.dehost_files(str(fq), run_extract=True).strip().split("\n")
    assert len(rows2) == 7
    # FASTA input has mean quality 0 and is never classified at the default --min_quality
    fa = tmp_path / "r.fasta"
    with open(fa, "w") as f:
        for i, s in enumerate(reads):
            f.write(">r%d\n%s\n" % (i, s.decode()))
    assert all(x.startswith("U\t") for x in oidx.dehost_files(str(fa)).strip().split("\n"))
    oidx.free()


def test_gamma_and_beta_densities_against_scipy(oracle_lib):
    """`charon classify` models reads with stats::dgamma / stats::dbeta (kthohr/stats 3.4.0, include/classify_stats.hpp:377-381).
    statslib is not in the image (parity unpinned): the oracle's float restatement is cross-checked against scipy.stats here, and
    the GPU kernel / the CLI are checked against the oracle in the -m gpu tests."""
    import scipy.stats as st
    xs = [1e-6, 0.001, 0.01, 0.05, 0.1, 0.15, 0.2, 0.37, 0.5, 0.75, 0.9, 0.999]
    for shape, scale in ((25.0, 0.02), (10.0, 0.005), (1.0, 0.3), (0.5, 2.0), (3.7, 0.11)):
        for x in xs:
            got, want = oracle_lib.density("gamma", x, shape, scale), st.gamma.pdf(float(np.float32(x)), shape, scale=float(np.float32(scale)))
            assert got == pytest.approx(want, rel=1e-4, abs=1e-38), (shape, scale, x)  # float log / lgamma / exp: a few 1e-5 in the far tails
    for a, b in ((6.0, 4.0), (6.0, 40.0), (1.0, 1.0), (0.5, 0.5), (2.5, 1.0), (1.0, 3.0)):
        for x in xs:
            got, want = oracle_lib.density("beta", x, a, b), st.beta.pdf(float(np.float32(x)), a, b)
            assert got == pytest.approx(want, rel=1e-4, abs=1e-38), (a, b, x)
    # boundaries of the support (statslib's limit values)
    assert oracle_lib.density("gamma", -0.1, 25, 0.02) == 0 and oracle_lib.density("gamma", 0.0, 25, 0.02) == 0
    assert oracle_lib.density("gamma", 0.0, 1.0, 0.25) == 4.0 and np.isinf(oracle_lib.density("gamma", 0.0, 0.5, 1.0))
    assert oracle_lib.density("beta", 0.0, 6, 4) == 0 and oracle_lib.density("beta", 1.0, 6, 4) == 0
    assert oracle_lib.density("beta", 0.0, 1.0, 3.0) == 3.0 and oracle_lib.density("beta", 1.0, 2.5, 1.0) == 2.5
    assert oracle_lib.density("beta", 1.5, 6, 4) == 0 and np.isnan(oracle_lib.density("beta", float("nan"), 6, 4))
    # method-of-moments fits (include/classify_stats.hpp:127-142,171-191)
    r = np.random.default_rng(3)
    data = r.gamma(9.0, 0.03, 4000).astype(np.float32)
    shape, loc, scale = oracle_lib.fit("gamma", data, (25.0, 0.0, 0.02))
    mu = float(np.mean(data.astype(np.float64)))
    s = np.log(mu) - float(np.mean(np.log(data).astype(np.float64)))
    assert shape == pytest.approx((3 - s + np.sqrt((s - 3) ** 2 + 24 * s)) / (12 * s), rel=1e-5) and scale == pytest.approx(mu / shape, rel=1e-5) and loc == 0
    assert 7.5 < shape < 10.5
    _, loc2, _ = oracle_lib.fit("gamma", np.zeros(0, np.float32), (10.0, 0.0, 0.005), loc_only=True)
    assert loc2 == pytest.approx(-0.05)  # what force_ready does to the default neg distribution of a run without --extract
    d2 = r.beta(5.0, 30.0, 4000).astype(np.float32)
    alpha, beta, _ = oracle_lib.fit("beta", d2, (6.0, 40.0))
    m, v = float(np.mean(d2.astype(np.float64))), float(np.var(d2.astype(np.float64), ddof=1))
    assert alpha == pytest.approx(m * (m * (1 - m) / v - 1), rel=1e-3) and beta == pytest.approx((1 - m) * (m * (1 - m) / v - 1), rel=1e-3)
