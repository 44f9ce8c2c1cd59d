"""GPU tests (-m gpu) of the drop-in boundary: the C++14 `charon dehost` front end (charon_amd/bin/charon, linked only
against the C ABI) must print the same TSV as the CPU oracle's dehost -- same rows in the same order (including the
dropped-first-read quirk), every column textually identical except the probability, which may differ by <= 1e-6."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from oracle import pyref
from tests import util

pytestmark = pytest.mark.gpu
G = os.path.join(util.ROOT, "tests", "golden")
EXE = os.path.join(util.ROOT, "charon_amd", "bin", "charon")


def run_cli(args, cwd, env_extra=None, sub="dehost"):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([EXE, sub] + args + ["--log", os.path.join(cwd, "charon.log")], cwd=cwd, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def assert_same_tsv(got, want, prob_tol=1e-6):
    g, w = got.strip("\n").split("\n"), want.strip("\n").split("\n")
    assert len(g) == len(w), (len(g), len(w))
    for a, b in zip(g, w):
        if a == b:
            continue
        fa, fb = a.split("\t"), b.split("\t")
        assert fa[:8] == fb[:8], (a, b)
        da, db = fa[8].strip().split(" "), fb[8].strip().split(" ")
        assert len(da) == len(db)
        for x, y in zip(da, db):
            xs, ys = x.split(":"), y.split(":")
            assert xs[:4] == ys[:4], (a, b)
            if xs[4] != ys[4]:
                # the SIGN of a NaN probability (reads with no k-mer) is compiler codegen noise (x86 propagates the sign
                # of whichever operand the compiler put first); NaN-ness itself must agree
                if "nan" in xs[4] or "nan" in ys[4]:
                    assert "nan" in xs[4] and "nan" in ys[4], (a, b)
                else:
                    assert abs(float(xs[4]) - float(ys[4])) <= prob_tol, (a, b)


def test_cli_golden_cfg1(tmp_path):
    assert os.path.exists(EXE), "host front end not built"
    fq = os.path.join(G, "cfg1_reads.fastq.gz")
    rc, out, err = run_cli(["--db", os.path.join(G, "cfg1.idx"), fq], str(tmp_path))
    assert rc == 0, err
    assert_same_tsv(out, open(os.path.join(G, "cfg1_expected.tsv")).read())
    assert out.split("\n")[0].split("\t")[1] == "r1"  # r0 is dropped (no --extract)
    # loader self-check (v): the index records the reference FASTA paths; when they are readable their minimisers are probed
    log = open(tmp_path / "charon.log").read()
    if all(os.path.exists(p) for p in ("/root/repo/tests/golden/my.fasta", "/root/repo/tests/golden/cfg1_host.fasta")):
        assert "self-check: minimisers of 2 reference file(s) all found" in log
    assert "self-check FAILED" not in log and "self-check FAILED" not in err
    # small GPU batches, more host threads: identical rows
    rc, out2, err = run_cli(["--db", os.path.join(G, "cfg1.idx"), "-t", "4", fq], str(tmp_path), {"CHARON_BATCH_READS": "37"})
    assert rc == 0 and out2 == out
    # the same reads as a one-stream .gz cut into chunks (this build's inflate, several threads), as BGZF, through zlib, and as plain text
    import subprocess, sys
    data = gzip.decompress(open(fq, "rb").read())
    (tmp_path / "plain.fastq").write_bytes(data)
    subprocess.run([sys.executable, os.path.join(util.ROOT, "tools", "make_bgzf.py"), str(tmp_path / "plain.fastq"), str(tmp_path / "b.fastq.gz"), "6", "2"], check=True)
    (tmp_path / "l1.fastq.gz").write_bytes(gzip.compress(data, 1))
    for f, env in ((fq, {"CHARON_INFLATE_CHUNK": "2048"}), (str(tmp_path / "l1.fastq.gz"), {"CHARON_INFLATE_CHUNK": "4096"}),
                   (str(tmp_path / "b.fastq.gz"), {}), (fq, {"CHARON_ZLIB_INFLATE": "1"}), (str(tmp_path / "plain.fastq"), {})):
        rc, o, err = run_cli(["--db", os.path.join(G, "cfg1.idx"), "-t", "4", f], str(tmp_path), env)
        assert rc == 0 and o == out, (f, env, err)
    # --extract drives the training cache (num_reads_to_fit 20 -> models retrain, cached reads re-classified)
    rc, out3, err = run_cli(["--db", os.path.join(G, "cfg1.idx"), "--extract", "microbial", "--num_reads_to_fit", "20", fq], str(tmp_path),
                            {"CHARON_BATCH_READS": "64"})
    assert rc == 0, err
    assert_same_tsv(out3, open(os.path.join(G, "cfg1_expected_extract.tsv")).read())


def test_cli_paired_fasta_and_thresholds(tmp_path, oracle_lib):
    r = util.rng(21)
    gs = [util.random_seq(r, 6000) for _ in range(3)]
    for i, g in enumerate(gs):
        with open(tmp_path / ("g%d.fa" % i), "w") as f:
            f.write(">g%d\n%s\n" % (i, g.decode()))
    oidx = oracle_lib.Index.from_fasta([(str(tmp_path / "g0.fa"), "human"), (str(tmp_path / "g1.fa"), "bacteria"),
                                        (str(tmp_path / "g2.fa"), "human")], ["bacteria", "human"])
    oidx.store(str(tmp_path / "p.idx"))
    m1 = util.sample_reads(r, gs, 300, (100, 250), sub_rate=0.02)
    m2 = util.sample_reads(r, gs, 300, (100, 250), sub_rate=0.02)
    m1[5] = m1[5][:60] + b"NNNRY" + m1[5][65:]
    for name, mates, tag in (("r_1.fastq", m1, "/1"), ("r_2.fastq", m2, "/2")):
        with open(tmp_path / name, "w") as f:
            for i, s in enumerate(mates):
                q = "".join(chr(33 + int(x)) for x in r.integers(5, 41, len(s)))
                f.write("@read%d%s\n%s\n+\n%s\n" % (i, tag, s.decode(), q))
    want = oidx.dehost_files(str(tmp_path / "r_1.fastq"), str(tmp_path / "r_2.fastq"))
    # --db carries CLI::ExistingPath (src/dehost_main.cpp:226-229) and ".idx" is appended afterwards (:489-491): a bare
    # prefix is rejected by the parser unless something exists at that path
    rc, out, err = run_cli(["--db", str(tmp_path / "p"), str(tmp_path / "r_1.fastq"), str(tmp_path / "r_2.fastq")], str(tmp_path))
    assert rc != 0 and "does not exist" in err
    open(tmp_path / "p", "w").close()
    rc, out, err = run_cli(["--db", str(tmp_path / "p"), str(tmp_path / "r_1.fastq"), str(tmp_path / "r_2.fastq")], str(tmp_path))
    assert rc == 0, err
    assert_same_tsv(out, want)
    assert sum(1 for x in out.split("\n") if x.startswith("C\t")) > 20
    # single-end, FASTA input (no qualities -> never classified), gz input, non-default thresholds
    with gzip.open(tmp_path / "s.fasta.gz", "wt") as f:
        for i, s in enumerate(m1):
            f.write(">s%d some description\n%s\n" % (i, s.decode()))
    want = oidx.dehost_files(str(tmp_path / "s.fasta.gz"))
    rc, out, err = run_cli(["--db", str(tmp_path / "p.idx"), str(tmp_path / "s.fasta.gz")], str(tmp_path))
    assert rc == 0, err
    assert_same_tsv(out, want)
    assert all(x.startswith("U\t") for x in out.strip().split("\n"))
    want = oidx.dehost_files(str(tmp_path / "r_1.fastq"), min_quality=0.0, confidence=200)
    rc, out, err = run_cli(["--db", str(tmp_path / "p.idx"), "--min_quality", "0", "--confidence", "200", str(tmp_path / "r_1.fastq")], str(tmp_path))
    assert rc == 0, err
    assert_same_tsv(out, want)
    oidx.free()


def test_cli_reader_formats(tmp_path, oracle_lib):
    """wrapped (multi-line) FASTA and FASTQ, CRLF line ends, blank lines, a zero-length read, lower case + IUPAC codes:
    the block reader must see the same records as the oracle's reader"""
    r = util.rng(33)
    gs = [util.random_seq(r, 5000), util.random_seq(r, 5000)]
    oidx = util.build_oracle_index(oracle_lib, [[gs[0]], [gs[1]]], [0, 1], ["host", "microbial"])
    oidx.compress()
    oidx.store(str(tmp_path / "f.idx"))
    reads = util.sample_reads(r, gs, 120, (30, 900), sub_rate=0.02)
    reads[3] = reads[3][:100].lower() + b"RYKMSWN" + reads[3][107:]
    reads[9] = b""
    wrap = lambda s, w: "\n".join(s[i:i + w] for i in range(0, len(s), w)) if s else ""
    with open(tmp_path / "w.fastq", "w", newline="") as f:
        for i, s in enumerate(reads):
            q = "".join(chr(33 + int(x)) for x in r.integers(20, 41, len(s)))
            eol = "\r\n" if i % 3 == 0 else "\n"
            f.write(("@w%d desc" + eol + "%s" + eol + "+" + eol + "%s" + eol) % (i, wrap(s.decode(), 70).replace("\n", eol), wrap(q, 70).replace("\n", eol)))
            if i % 7 == 0:
                f.write(eol)
    with open(tmp_path / "w.fa", "w") as f:
        for i, s in enumerate(reads):
            f.write(">w%d\n%s\n" % (i, wrap(s.decode(), 60)))
    for name, extra in (("w.fastq", []), ("w.fa", ["--min_quality", "0"])):
        mq = 0.0 if extra else 15.0
        want = oidx.dehost_files(str(tmp_path / name), min_quality=mq)
        for env in ({}, {"CHARON_BATCH_READS": "17"}):
            rc, out, err = run_cli(["--db", str(tmp_path / "f.idx")] + extra + [str(tmp_path / name)], str(tmp_path), env)
            assert rc == 0, err
            assert_same_tsv(out, want)
    # the same file as .bz2 (this build's own block-parallel decoder): same rows
    import bz2
    (tmp_path / "z.fastq.bz2").write_bytes(bz2.compress(open(tmp_path / "w.fastq", "rb").read(), 1))
    rc, out, err = run_cli(["--db", str(tmp_path / "f.idx"), "-t", "4", str(tmp_path / "z.fastq.bz2")], str(tmp_path))
    assert rc == 0, err
    assert_same_tsv(out, oidx.dehost_files(str(tmp_path / "w.fastq"), min_quality=15.0))
    # an illegal sequence character is a parse error (non-zero exit), as in seqan3
    with open(tmp_path / "bad.fastq", "w") as f:
        f.write("@x\nACGTXACGT\n+\nIIIIIIIII\n")
    rc, out, err = run_cli(["--db", str(tmp_path / "f.idx"), str(tmp_path / "bad.fastq")], str(tmp_path))
    assert rc != 0
    oidx.free()


def _read_fastq_gz(path):
    recs = []
    with gzip.open(path, "rt") as f:
        lines = f.read().split("\n")
    i = 0
    while i + 3 < len(lines) + 1 and i < len(lines) and lines[i]:
        recs.append((lines[i][1:], lines[i + 1], lines[i + 3]))
        i += 4
    return recs


def test_cli_extract_writes_called_reads(tmp_path):
    """--extract: records of reads called as the category go to <prefix>_<category><ext>.gz in classification order
    (src/dehost_main.cpp:515-536, include/result.hpp:118-128); sequences come out as dna5 letters"""
    fq = os.path.join(G, "cfg1_reads.fastq.gz")
    originals = {r[0].split(" ")[0]: r for r in _read_fastq_gz(fq)}
    pre = str(tmp_path / "out")
    rc, out, err = run_cli(["--db", os.path.join(G, "cfg1.idx"), "--extract", "microbial", "-p", pre, "--num_reads_to_fit", "20", fq], str(tmp_path))
    assert rc == 0, err
    assert_same_tsv(out, open(os.path.join(G, "cfg1_expected_extract.tsv")).read())
    want_ids = [x.split("\t")[1] for x in out.strip().split("\n") if x.split("\t")[2] == "microbial"]
    got = _read_fastq_gz(pre + "_microbial.fastq.gz")
    assert [g[0].split(" ")[0] for g in got] == want_ids and len(got) > 20
    for gid, seq, qual in got:
        o = originals[gid.split(" ")[0]]
        assert gid == o[0] and qual == o[2]
        assert seq == "".join(c if c in "ACGT" else "N" for c in o[1].upper())
    assert not os.path.exists(pre + "_host.fastq.gz")
    # all categories, default prefix "charon" in the working directory
    rc, out, err = run_cli(["--db", os.path.join(G, "cfg1.idx"), "-e", "all", fq], str(tmp_path))
    assert rc == 0, err
    n_m = len(_read_fastq_gz(str(tmp_path / "charon_microbial.fastq.gz")))
    n_h = len(_read_fastq_gz(str(tmp_path / "charon_host.fastq.gz")))
    rows = out.strip().split("\n")
    assert n_m == sum(1 for x in rows if x.split("\t")[2] == "microbial") and n_h == sum(1 for x in rows if x.split("\t")[2] == "host")
    # an unknown category is logged, nothing runs, exit status 0
    rc, out, err = run_cli(["--db", os.path.join(G, "cfg1.idx"), "-e", "fungi", fq], str(tmp_path))
    assert rc == 0 and out == "" and "Cannot extract fungi" in open(tmp_path / "charon.log").read()


def test_cli_self_check_catches_foreign_index(tmp_path, oracle_lib):
    """if the reference FASTA named in the index does not hash into its bin, the loader says so loudly (hard part H6)"""
    r = util.rng(44)
    for i in range(2):
        with open(tmp_path / ("g%d.fa" % i), "w") as f:
            f.write(">g%d\n%s\n" % (i, util.random_seq(r, 4000).decode()))
    oidx = oracle_lib.Index.from_fasta([(str(tmp_path / "g0.fa"), "host"), (str(tmp_path / "g1.fa"), "other")], ["host", "other"])
    oidx.store(str(tmp_path / "s.idx"))
    oidx.free()
    with open(tmp_path / "g0.fa", "w") as f:  # the file on disk no longer is what was indexed
        f.write(">g0\n%s\n" % util.random_seq(r, 4000).decode())
    with open(tmp_path / "q.fastq", "w") as f:
        f.write("@a\n%s\n+\n%s\n" % ("ACGT" * 50, "I" * 200))
    rc, out, err = run_cli(["--db", str(tmp_path / "s.idx"), str(tmp_path / "q.fastq")], str(tmp_path))
    assert rc == 0 and "self-check FAILED" in err
    assert "self-check FAILED" in open(tmp_path / "charon.log").read()


def test_cli_index_builds_the_same_index_as_the_oracle(tmp_path, oracle_lib):
    """`charon index` (GPU minimisers of chunked references + device emplace + EF writer) must produce the same IBF words
    and metadata as the oracle's sequential builder, and the file must load everywhere"""
    r = util.rng(55)
    files = []
    for i, (nrec, L) in enumerate(((1, 30000), (3, 9000), (2, 50), (1, 5000), (2, 60))):
        path = tmp_path / ("ref%d.fa" % i)
        with open(path, "w") as f:
            for j in range(nrec):
                s = util.random_seq(r, L).decode()
                if i == 1 and j == 1:
                    s = s[:4000] + "N" * 30 + s[4030:6000].lower() + "AC" * 200 + s[6400:]
                f.write(">rec%d_%d\n" % (i, j))
                f.write("\n".join(s[k:k + 70] for k in range(0, len(s), 70)) + "\n")
        files.append(str(path))
    cats = ["microbial", "human", "microbial", "human", "microbial"]
    tab = tmp_path / "in.tab"
    with open(tab, "w") as f:
        for p, c in zip(files, cats):
            f.write("%s\t%s\n" % (p, c))
    p = subprocess.run([EXE, "index", "-p", str(tmp_path / "built"), "--log", str(tmp_path / "i.log"), str(tab)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    got = oracle_lib.Index.load(str(tmp_path / "built.idx"))
    want = oracle_lib.Index.from_fasta(list(zip(files, cats)), got.categories)
    assert sorted(got.categories) == ["human", "microbial"]
    assert (got.k, got.w, got.bins, got.bin_size, got.hash_funs) == (want.k, want.w, want.bins, want.bin_size, want.hash_funs)
    assert list(got.bin_to_cat) == list(want.bin_to_cat)
    assert np.array_equal(got.words(), want.words())
    # the file closes with the sd_vector's two select_support_mcl blocks (what upstream's archive(ibf_) stores): the oracle's loader answers
    # select from them, and they are, byte for byte, what the oracle's statement-by-statement restatement of sdsl writes for this m_high
    L = oracle_lib.lib()
    assert L.orc_sd_has_select(got.h) == 1
    meta = pyref.read_index(str(tmp_path / "built.idx"))
    hb = pyref.bit_array(meta["high"], meta["high_bits"])
    for b, d in ((1, meta["select1"]), (0, meta["select0"])):
        pos = np.flatnonzero(hb == b)
        assert d["arg_cnt"] == len(pos) == L.orc_sd_select_args(got.h, b)
        for i in (1, 2, 64, 65, 4096, 4097, len(pos) // 2, len(pos)):
            assert pyref.select_from_blocks(d, hb, b, i) == int(pos[i - 1]) == L.orc_sd_select(got.h, b, i)
    words = np.packbits(np.concatenate([hb, np.zeros((-len(hb)) % 64, np.uint8)]), bitorder="little").view(np.uint64)
    tail = oracle_lib.select_blocks(words, len(hb), str(tmp_path / "tail.bin"))
    assert open(tmp_path / "built.idx", "rb").read().endswith(tail) and len(tail) > 100
    # chromosome-sized records are processed in overlapping pieces; force tiny pieces to exercise that path -- and build on five
    # threads (several-thread sort of the minimiser sets, Elias-Fano arrays written by all threads at once): the same file, byte for byte
    p2 = subprocess.run([EXE, "index", "-t", "5", "-p", str(tmp_path / "pieces"), "--log", str(tmp_path / "i.log"), str(tab)], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, env=dict(os.environ, CHARON_INDEX_PIECE="8192", CHARON_PSORT_MIN="16"))
    assert p2.returncode == 0, p2.stderr.decode()
    assert open(tmp_path / "pieces.idx", "rb").read() == open(tmp_path / "built.idx", "rb").read()
    # and it is usable: dehost some reads with it, same TSV as the oracle on the oracle-built twin
    reads = util.sample_reads(r, [open(f).read().split("\n", 1)[1].replace("\n", "").replace(">", "").encode()[:20000] for f in files[:2]], 60, (200, 900))
    with open(tmp_path / "q.fastq", "w") as f:
        for i, s in enumerate(reads):
            s = s.decode().upper()
            s = "".join(c if c in "ACGTN" else "A" for c in s)
            f.write("@q%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    rc, out, err = run_cli(["--db", str(tmp_path / "built.idx"), str(tmp_path / "q.fastq")], str(tmp_path))
    assert rc == 0, err
    assert_same_tsv(out, want.dehost_files(str(tmp_path / "q.fastq")))
    # --optimize merges small bins of one category; every reference must still hit its (new) bin completely (self-check v)
    p = subprocess.run([EXE, "index", "--optimize", "-p", str(tmp_path / "opt"), "--log", str(tmp_path / "i.log"), str(tab)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    opt = oracle_lib.Index.load(str(tmp_path / "opt.idx"))
    assert opt.bins == 4 and opt.ncat == 2  # the two tiny microbial files share a bucket (src/index_main.cpp:193-201)
    rc, out, err = run_cli(["--db", str(tmp_path / "opt.idx"), str(tmp_path / "q.fastq")], str(tmp_path))
    assert rc == 0 and "self-check FAILED" not in err
    assert "reference file(s) all found" in open(tmp_path / "charon.log").read()
    got.free(); want.free(); opt.free()


def test_cli_errors(tmp_path):
    fq = os.path.join(G, "cfg1_reads.fastq.gz")
    rc, out, err = run_cli(["--db", os.path.join(G, "cfg1.idx"), "--threads", "256", fq], str(tmp_path))
    assert rc != 0 and out == ""
    rc, out, err = run_cli(["--db", os.path.join(G, "nope.idx"), fq], str(tmp_path))
    assert rc != 0
    # rejected --dist is logged and the exit status stays 0 (the reference's callback drops the return value)
    rc, out, err = run_cli(["--db", os.path.join(G, "cfg1.idx"), "--dist", "weibull", fq], str(tmp_path))
    assert rc == 0 and out == ""
    assert "Supported distributions" in open(tmp_path / "charon.log").read()


def test_cli_rejects_corrupted_index_payload(tmp_path):
    """the sd_vector payload is decoded on the device (chn_index_decode_ef); a damaged m_high / m_low must be reported by the
    loader's self-checks (ones in m_high vs elements of m_low, bits beyond m_size or in technical bins >= num_bins, distinct bits
    after decode), never classified with"""
    fq = os.path.join(G, "cfg1_reads.fastq.gz")
    src = open(os.path.join(G, "cfg1.idx"), "rb").read()
    n = os.path.getsize(os.path.join(G, "cfg1_no_select.idx"))  # the same file up to the end of m_high
    outcomes = []
    for where, mask in ((n - 9, 0xFF), (n - 2000, 0x10), (n // 2, 0x01), (n // 2 + 777, 0x80)):
        b = bytearray(src)
        b[where] ^= mask
        p = tmp_path / "bad.idx"
        p.write_bytes(bytes(b))
        rc, out, err = run_cli(["--db", str(p), fq], str(tmp_path))
        outcomes.append((rc, err.strip().split("\n")[-1] if err.strip() else ""))
    # every flip changes the decoded bit set; the damage is either caught by a structural check (rc != 0) or - when the flip moved
    # one bit to another legal position - by the reference-file self-check warning in the log
    assert any(rc != 0 for rc, _ in outcomes), outcomes
    for rc, msg in outcomes:
        assert rc != 0 or "self-check" in open(tmp_path / "charon.log").read(), outcomes


def test_cli_index_select_blocks_checked_on_load(tmp_path):
    """a file that carries the sd_vector's two select_support_mcl blocks (upstream's layout) has them verified against
    m_high on load; damage there is an error with a file offset; a file of an earlier build of this program (ends behind m_high) still loads"""
    fq = os.path.join(G, "cfg1_reads.fastq.gz")
    want = open(os.path.join(G, "cfg1_expected.tsv")).read()
    rc, out, err = run_cli(["--db", os.path.join(G, "cfg1.idx"), fq], str(tmp_path))
    assert rc == 0, err
    assert_same_tsv(out, want)
    assert "select_support_mcl blocks answer select exactly as m_high does" in open(tmp_path / "charon.log").read()
    rc, out, err = run_cli(["--db", os.path.join(G, "cfg1_no_select.idx"), fq], str(tmp_path))
    assert rc == 0, err
    assert_same_tsv(out, want)
    assert "ends behind m_high" in open(tmp_path / "charon.log").read()
    # the loader's slices (an index of the published size goes over in some 600 of them, staged by the reader's threads while the device decodes
    # the slice before): the fixture in slices of 1, 7 and 64 words of m_high -- the running count of ones and the m_low offsets across slices
    for words in ("1", "7", "64"):
        rc, out, err = run_cli(["--db", os.path.join(G, "cfg1.idx"), fq], str(tmp_path), {"CHARON_INDEX_SLICE": words})
        assert rc == 0, (words, err)
        assert_same_tsv(out, want)
    src = open(os.path.join(G, "cfg1.idx"), "rb").read()
    n0 = os.path.getsize(os.path.join(G, "cfg1_no_select.idx"))
    for what, buf in (("arg_cnt", src[:n0] + bytes([src[n0] ^ 1]) + src[n0 + 1:]), ("stored answer", src[:n0 + 41] + bytes([src[n0 + 41] ^ 4]) + src[n0 + 42:]),
                      ("truncated", src[:-5]), ("trailing bytes", src + b"\0\0\0")):
        p = tmp_path / "bad.idx"
        p.write_bytes(buf)
        rc, out, err = run_cli(["--db", str(p), fq], str(tmp_path))
        assert rc != 0 and out == "" and "select_support_mcl" in err and "file offset" in err, (what, err)


def test_cli_long_reads_compression_column(tmp_path, oracle_lib):
    """reads longer than zlib's 32 KiB window: the gzip-size emulator has to reproduce zlib's window slide; the TSV (incl. the
    compression column, which the oracle computes with zlib itself) must still be identical, in both modes of the front end"""
    r = util.rng(808)
    gs = [util.random_seq(r, 200000), util.random_seq(r, 200000)]
    oidx = util.build_oracle_index(oracle_lib, [[gs[0]], [gs[1]]], [0, 1], ["host", "microbial"])
    oidx.compress()
    oidx.store(str(tmp_path / "l.idx"))
    reads = [util.mutate(r, gs[0][100:70100], 0.05), util.mutate(r, gs[1][:150000], 0.08), gs[0][5000:10000], b"ACGT" * 20000,
             util.mutate(r, gs[1][1000:66300], 0.02), gs[0][:65274] + b"N" * 40]
    with open(tmp_path / "long.fastq", "w") as f:
        for i, s in enumerate(reads):
            f.write("@L%d\n%s\n+\n%s\n" % (i, s.decode(), "I" * len(s)))
    want = oidx.dehost_files(str(tmp_path / "long.fastq"))
    for env in ({}, {"CHARON_ZLIB_ONLY": "1"}):
        rc, out, err = run_cli(["--db", str(tmp_path / "l.idx"), "-t", "4", str(tmp_path / "long.fastq")], str(tmp_path), env)
        assert rc == 0, err
        assert_same_tsv(out, want)
    oidx.free()


def test_cli_classify_gamma_beta(tmp_path, oracle_lib):
    """`charon classify` (src/classify_main.cpp): call_category for single-end reads too, beta (default) / gamma models, its own
    threshold defaults incl. min_compression 0.15 (so the gzip column matters), dropped first read; --extract trains the models
    (method-of-moments fits, include/classify_stats.hpp:127-142,171-191) and re-classifies the cached reads; `charon dehost
    --dist gamma` shares the machinery.  TSV == the oracle's classify."""
    r = util.rng(33)
    gs = [util.random_seq(r, 8000) for _ in range(3)]
    for i, g in enumerate(gs):
        with open(tmp_path / ("g%d.fa" % i), "w") as f:
            f.write(">g%d\n%s\n" % (i, g.decode()))
    oidx = oracle_lib.Index.from_fasta([(str(tmp_path / "g0.fa"), "human"), (str(tmp_path / "g1.fa"), "bacteria"),
                                        (str(tmp_path / "g2.fa"), "virus")], ["bacteria", "human", "virus"])
    oidx.store(str(tmp_path / "c.idx"))
    reads = util.sample_reads(r, gs, 400, (120, 900), sub_rate=0.03, random_fraction=0.15) + [b"ACGT" * 100, b"A" * 400]
    with open(tmp_path / "r.fastq", "w") as f:
        for i, s in enumerate(reads):
            q = "".join(chr(33 + int(x)) for x in r.integers(8, 41, len(s)))
            f.write("@read%d extra\n%s\n+\n%s\n" % (i, s.decode(), q))
    fq, db = str(tmp_path / "r.fastq"), str(tmp_path / "c.idx")
    for dist in ("beta", "gamma"):
        want = oidx.classify_files(fq, dist=dist)
        rc, out, err = run_cli(["--db", db] + ([] if dist == "beta" else ["--dist", dist]) + [fq], str(tmp_path), sub="classify")
        assert rc == 0, err
        assert_same_tsv(out, want)
        assert len(out.strip().split("\n")) == len(reads) - 1  # the first read is dropped here as well
        calls = {ln.split("\t")[2] for ln in out.strip().split("\n")}
        assert {"bacteria", "human", "virus"} <= calls
    # low-complexity reads fail classify's min_compression 0.15 gate
    last = out.strip().split("\n")[-1].split("\t")
    assert last[0] == "U" and float(last[7]) < 0.15
    # training: --extract with a small num_reads_to_fit (fits happen, cached reads are re-classified), small GPU batches
    for dist in ("beta", "gamma"):
        want = oidx.classify_files(fq, run_extract=True, num_reads_to_fit=25, dist=dist)
        rc, out, err = run_cli(["--db", db, "-d", dist, "--extract", "all", "--num_reads_to_fit", "25", "-p", str(tmp_path / ("x" + dist)), fq],
                               str(tmp_path), {"CHARON_BATCH_READS": "48"}, sub="classify")
        assert rc == 0, err
        assert_same_tsv(out, want)
        assert os.path.exists(tmp_path / ("x%s_bacteria.fastq.gz" % dist))
    # kde is not a classify distribution (src/classify_main.cpp:338-341): logged, no rows, exit status 0
    rc, out, err = run_cli(["--db", db, "-d", "kde", fq], str(tmp_path), sub="classify")
    assert rc == 0 and out == "" and "Supported distributions are [gamma , beta]" in open(tmp_path / "charon.log").read()
    # paired classify
    m2 = util.sample_reads(r, gs, 120, (100, 250), sub_rate=0.02)
    m1 = util.sample_reads(r, gs, 120, (100, 250), sub_rate=0.02)
    for name, mates, tag in (("p_1.fastq", m1, "/1"), ("p_2.fastq", m2, "/2")):
        with open(tmp_path / name, "w") as f:
            for i, s in enumerate(mates):
                f.write("@pr%d%s\n%s\n+\n%s\n" % (i, tag, s.decode(), "I" * len(s)))
    want = oidx.classify_files(str(tmp_path / "p_1.fastq"), str(tmp_path / "p_2.fastq"))
    rc, out, err = run_cli(["--db", db, str(tmp_path / "p_1.fastq"), str(tmp_path / "p_2.fastq")], str(tmp_path), sub="classify")
    assert rc == 0, err
    assert_same_tsv(out, want)
    oidx.free()
    # dehost accepts gamma / beta as well (src/dehost_main.cpp:538-541)
    h = [util.random_seq(r, 8000) for _ in range(2)]
    for i, g in enumerate(h):
        with open(tmp_path / ("h%d.fa" % i), "w") as f:
            f.write(">h%d\n%s\n" % (i, g.decode()))
    oidx = oracle_lib.Index.from_fasta([(str(tmp_path / "h0.fa"), "human"), (str(tmp_path / "h1.fa"), "bacteria")], ["bacteria", "human"])
    oidx.store(str(tmp_path / "d.idx"))
    reads = util.sample_reads(r, h, 200, (150, 900), sub_rate=0.03)
    with open(tmp_path / "d.fastq", "w") as f:
        for i, s in enumerate(reads):
            f.write("@d%d\n%s\n+\n%s\n" % (i, s.decode(), "I" * len(s)))
    for dist in ("gamma", "beta"):
        want = oidx.dehost_files(str(tmp_path / "d.fastq"), dist=dist)
        rc, out, err = run_cli(["--db", str(tmp_path / "d.idx"), "--dist", dist, str(tmp_path / "d.fastq")], str(tmp_path))
        assert rc == 0, err
        assert_same_tsv(out, want)
    oidx.free()


def test_cli_gzip_column_from_device_tallies(tmp_path, oracle_lib):
    """the `compression` column (src/utils.cpp:114-124): by default the deflate pass runs on the GPU (k_gzip_tally: zlib's level-6
    deflate_slow as code-frequency tallies, sized on the host by _tr_flush_block's arithmetic); it must print exactly what the
    host emulator (CHARON_GZIP_ON_HOST=1) and zlib itself (CHARON_ZLIB_ONLY=1) print, for every shape of read: random, N-rich,
    tandem repeats, internal copies (long matches, lazy evaluation), homopolymers, short reads, reads beyond the device's length
    limit and beyond one deflate block (both sized on the host), pairs (the two mates are compressed as ONE string)."""
    r = np.random.default_rng(55)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    g0 = util.random_seq(util.rng(1), 6000)
    for i, g in enumerate((g0, util.random_seq(util.rng(2), 6000))):
        with open(tmp_path / ("g%d.fa" % i), "w") as f:
            f.write(">g%d\n%s\n" % (i, g.decode()))
    oidx = oracle_lib.Index.from_fasta([(str(tmp_path / "g0.fa"), "human"), (str(tmp_path / "g1.fa"), "microbial")], ["microbial", "human"])
    oidx.store(str(tmp_path / "z.idx"))
    oidx.free()
    recs = []
    lengths = [1, 2, 3, 4, 5, 7, 19, 41, 150, 300, 1000, 5000, 5000, 5000, 9000, 16383, 16384, 16385, 20000, 33000, 61440, 61441, 70000]
    for i in range(420):
        L = int(lengths[i % len(lengths)]) if i % 3 else int(r.integers(1, 4000))
        s = acgt[r.integers(0, 4, L)].copy()
        kind = i % 7
        if kind == 1:
            s[r.random(L) < 0.04] = ord("N")
        elif kind == 2 and L > 20:
            s = np.resize(acgt[r.integers(0, 4, int(r.integers(1, 9)))], L)          # tandem repeat
        elif kind == 3 and L > 200:
            for _ in range(12):                                                          # internal copies
                a, b, ln = int(r.integers(0, L)), int(r.integers(0, L)), int(r.integers(10, 600))
                ln = min(ln, L - a, L - b)
                s[b:b + ln] = s[a:a + ln].copy()
        elif kind == 4:
            s[:] = acgt[int(r.integers(0, 4))]                                           # homopolymer
        elif kind == 5 and L > 40:
            s[L // 2:] = s[:L - L // 2].copy()                                           # one long repeat
        elif kind == 6 and L > 3000:
            s = np.resize(acgt[r.integers(0, 4, 3)], L)                                  # period-3 repeat: > 16 383 symbols? no: long matches
        recs.append(bytes(s))
    with open(tmp_path / "z.fasta", "w") as f:
        for i, s in enumerate(recs):
            f.write(">z%d\n%s\n" % (i, s.decode()))
    args = ["--db", str(tmp_path / "z.idx"), "--min_quality", "0", str(tmp_path / "z.fasta")]
    outs = {}
    for mode, env in (("gpu", {}), ("gpu16k", {"CHARON_GZIP_GPU_MAX": "16384"}), ("host", {"CHARON_GZIP_ON_HOST": "1"}), ("zlib", {"CHARON_ZLIB_ONLY": "1"})):
        rc, out, err = run_cli(args, str(tmp_path), dict(env, CHARON_BATCH_READS="100"))
        assert rc == 0, err
        outs[mode] = out
    assert outs["gpu"] == outs["zlib"] == outs["host"] == outs["gpu16k"]
    # ... and the device really did the work: everything up to 61 440 letters (-t 1: long reads go to the device too) that fits one deflate
    # block (the last run was the zlib one, so look at the log of a fresh default run)
    rc, out, err = run_cli(args, str(tmp_path))
    import re
    mm = re.search(r"gzip column: (\d+) reads sized from device deflate tallies", open(tmp_path / "charon.log").read())
    n_dev = sum(1 for s in recs[1:] if len(s) <= 61440)  # (the dropped first read is tallied too, but harmlessly)
    assert mm and n_dev - 12 <= int(mm.group(1)) <= n_dev + 1  # (a few long low-match reads need a second deflate block: host)
    assert len(outs["gpu"].strip().split("\n")) == len(recs) - 1
    # classify gates on the column (min_compression 0.15): low-complexity reads lose their call in every mode alike
    # pairs: both mates in one gzip member
    m1, m2 = recs[10:130], recs[200:320]
    for name, mates, tag in (("q_1.fasta", m1, "/1"), ("q_2.fasta", m2, "/2")):
        with open(tmp_path / name, "w") as f:
            for i, s in enumerate(mates):
                f.write(">q%d%s\n%s\n" % (i, tag, s.decode()))
    pargs = ["--db", str(tmp_path / "z.idx"), "--min_quality", "0", str(tmp_path / "q_1.fasta"), str(tmp_path / "q_2.fasta")]
    a = run_cli(pargs, str(tmp_path), sub="classify")
    b = run_cli(pargs, str(tmp_path), {"CHARON_ZLIB_ONLY": "1"}, sub="classify")
    assert a[0] == 0 and b[0] == 0 and a[1] == b[1] and a[1].count("\n") == len(m1) - 1
    assert any(ln.startswith("U") for ln in a[1].split("\n")) and any(ln.startswith("C") for ln in a[1].split("\n"))
