"""CPU tests of the `charon` front end's command line (no GPU is touched: every case ends in the argument parser).
Flags, integer limits and exit codes follow src/main.cpp:49-65 and src/dehost_main.cpp:208-312 of the reference."""
import os
import subprocess

import pytest

from tests import util

EXE = os.path.join(util.ROOT, "charon_amd", "bin", "charon")
G = os.path.join(util.ROOT, "tests", "golden")


def run(args):
    p = subprocess.run([EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build()


def test_version_and_help():
    rc, out, _ = run(["--version"])
    assert rc == 0 and "charon" in out
    rc, out, _ = run(["dehost", "--help"])
    assert rc == 0
    for flag in ("--db", "--extract", "--prefix", "--chunk_size", "--lo_hi_threshold", "--num_reads_to_fit", "--dist", "--min_length",
                 "--min_quality", "--min_compression", "--confidence", "--host_unique_prop_lo_threshold", "--min_proportion_diff",
                 "--min_probability_diff", "--log", "--threads"):
        assert flag in out
    assert run([])[0] != 0 and run(["frobnicate"])[0] != 0


def test_parse_errors_exit_non_zero():
    fq, db = os.path.join(G, "cfg1_reads.fastq.gz"), os.path.join(G, "cfg1.idx")
    for args in (["dehost", fq],                                   # --db is required
                 ["dehost", "--db", db],                           # <fastaq> is required
                 ["dehost", "--db", db, "/no/such/file.fq"],       # ExistingFile
                 ["dehost", "--db", "/no/such/index", fq],         # ExistingPath
                 ["dehost", "--db", db, "--chunk_size", "256", fq],      # uint8_t
                 ["dehost", "--db", db, "-t", "300", fq],
                 ["dehost", "--db", db, "--confidence", "-1", fq],
                 ["dehost", "--db", db, "--num_reads_to_fit", "65536", fq],  # uint16_t
                 ["dehost", "--db", db, "--min_quality", "abc", fq],
                 ["dehost", "--db", db, "--frobnicate", fq],
                 ["dehost", "--db", db, "-p", G, fq],              # NonexistentPath
                 ["dehost", "--db", db, fq, fq, fq]):              # at most two read files
        rc, out, err = run(args)
        assert rc != 0 and out == "" and err, args


def _fnv(seq):
    h = 1469598103934665603
    for c in seq:
        h ^= ord(c if c in "ACGT" else "N")
        h = (h * 1099511628211) & ((1 << 64) - 1)
    return h


def test_block_reader_sees_what_was_written(tmp_path):
    """the CLI's block reader (hidden `_records` diagnostic, no GPU) on FASTA/FASTQ with wrapped lines, CRLF, blank lines,
    lower case / IUPAC letters, empty reads and gz, read through tiny blocks so that records straddle block boundaries"""
    import gzip
    import numpy as np
    r = np.random.default_rng(8)
    letters = np.array(list("ACGTacgtNRYKMSWn"))
    recs = []
    for i in range(300):
        L = int(r.choice([0, 1, 18, 19, 70, 71, 140, int(r.integers(1, 3000))]))
        seq = "".join(letters[r.integers(0, len(letters) if i % 5 == 0 else 4, L)])
        qual = "".join(chr(33 + int(x)) for x in r.integers(0, 42, L))
        recs.append(("read%d some description %d" % (i, i), seq, qual))
    wrap = lambda s, w: [s[k:k + w] for k in range(0, len(s), w)] or [""]

    def write(path, fastq, width, eol, gz):
        op = gzip.open if gz else open
        with op(path, "wt", newline="") as f:
            for i, (rid, seq, qual) in enumerate(recs):
                if fastq:
                    f.write("@" + rid + eol + eol.join(wrap(seq, width)) + eol + "+" + eol + eol.join(wrap(qual, width)) + eol)
                else:
                    f.write(">" + rid + eol + eol.join(wrap(seq, width)) + eol)
                if i % 11 == 0:
                    f.write(eol)
    want_fq = ["%s\t%d\t%d\t%d\t%d" % (rid, len(s), len(q), sum(ord(c) - 33 for c in q), _fnv(s.upper())) for rid, s, q in recs]
    want_fa = ["%s\t%d\t0\t0\t%d" % (rid, len(s), _fnv(s.upper())) for rid, s, q in recs]
    cases = [("a.fastq", True, 10 ** 9, "\n", False), ("b.fq", True, 60, "\r\n", False), ("c.fastq.gz", True, 80, "\n", True),
             ("d.fasta", False, 70, "\n", False), ("e.fa.gz", False, 10 ** 9, "\r\n", True)]
    for name, fastq, width, eol, gz in cases:
        path = str(tmp_path / name)
        write(path, fastq, width, eol, gz)
        for max_recs, max_bytes in ((1000, 1 << 20), (7, 300), (1, 64)):
            rc, out, err = run(["_records", path, str(max_recs), str(max_bytes)])
            assert rc == 0, err
            got = out.strip("\n").split("\n")
            assert got == (want_fq if fastq else want_fa), (name, max_recs, max_bytes)
    # an empty-sequence FASTA record is legal; an illegal letter is only detected when packing (dehost), not here


def test_gzip_size_emulator_equals_the_linked_zlib(tmp_path):
    """the `compression` column (src/utils.cpp:114-124) is computed by a size-only restatement of zlib's level-6 deflate
    (charon_amd/csrc/host/gzip_size.hpp); it must give exactly the size the linked zlib gives, record by record (hidden `_gzsize`
    diagnostic, no GPU).  tools/gzip_size_check.cpp is the larger sweep (480 k cases clean in round 1)."""
    import numpy as np
    r = np.random.default_rng(5)
    recs = []
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for i in range(600):
        L = int(r.choice([1, 2, 3, 4, 7, 19, 41, 150, 300, 1000, 5000, 5000, 5000, 20000, 59999, 60000, 65280, 70000, 140000]))  # the last three slide zlib's window
        if i % 7 == 0:
            L = int(r.integers(1, 3000))
        s = acgt[r.integers(0, 4, L)].copy()
        kind = i % 6
        if kind == 1:
            s[r.random(L) < 0.03] = ord("N")
        elif kind == 2 and L > 20:
            unit = acgt[r.integers(0, 4, int(r.integers(1, 9)))]
            s = np.resize(unit, L)
        elif kind == 3 and L > 200:
            for _ in range(10):
                a, b, ln = int(r.integers(0, L)), int(r.integers(0, L)), int(r.integers(10, 400))
                ln = min(ln, L - a, L - b)
                s[b:b + ln] = s[a:a + ln].copy()
        elif kind == 4:
            s[:] = acgt[int(r.integers(0, 4))]
        elif kind == 5 and L > 10:
            s = np.frombuffer(bytes(s).lower(), np.uint8).copy()
            s[3] = ord("r")  # IUPAC -> N
        recs.append(bytes(s))
    path = tmp_path / "z.fa"
    with open(path, "wb") as f:
        for i, s in enumerate(recs):
            f.write(b">s%d\n" % i + s + b"\n")
    rc, out, err = run(["_gzsize", str(path)])
    assert rc == 0, err
    lines = out.strip().split("\n")
    assert lines[0] == "selfcheck\t1"
    assert len(lines) == 1 + len(recs)
    for ln in lines[1:]:
        name, z, e = ln.split("\t")
        assert z == e, ln


def test_truncated_or_corrupt_gz_input_is_an_error(tmp_path):
    """a cut-off or CRC-corrupt .gz must fail loudly (the reference's seqan3 / zlib stream throws), never pass for a shorter file"""
    data = open(os.path.join(G, "cfg1_reads.fastq.gz"), "rb").read()
    rc, out, _ = run(["_records", os.path.join(G, "cfg1_reads.fastq.gz")])
    assert rc == 0 and out.count("\n") == 200
    cut = tmp_path / "cut.fastq.gz"
    cut.write_bytes(data[:len(data) * 2 // 3])
    rc, out, err = run(["_records", str(cut)])
    assert rc != 0 and "gzip read error" in err
    bad = bytearray(data)
    bad[-6] ^= 0x5A  # inside the CRC32 trailer
    crc = tmp_path / "crc.fastq.gz"
    crc.write_bytes(bytes(bad))
    rc, out, err = run(["_records", str(crc)])
    assert rc != 0 and "gzip read error" in err
    # the same records through the mapped-file path (plain text) and through zlib (CHARON_NO_MMAP) are identical
    import gzip
    plain = tmp_path / "plain.fastq"
    plain.write_bytes(gzip.decompress(data))
    a = subprocess.run([EXE, "_records", str(plain), "50", "4096"], stdout=subprocess.PIPE)
    b = subprocess.run([EXE, "_records", str(plain), "50", "4096"], stdout=subprocess.PIPE, env=dict(os.environ, CHARON_NO_MMAP="1"))
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and a.stdout.count(b"\n") == 200
    empty = tmp_path / "empty.fastq"
    empty.write_bytes(b"")
    assert run(["_records", str(empty)])[0] == 0


def _bgzf(data, block=65280, level=6, eof_marker=True):
    """BGZF writer (SAM specification 4.1): gzip members of at most 64 KiB with their size in a 'BC' extra subfield"""
    import struct, zlib
    out = bytearray()
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof_marker else [])
    for c in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        z = co.compress(c) + co.flush()
        bsize = 12 + 6 + len(z) + 8 - 1
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
        out += z + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c))
    return bytes(out)


def test_bgzf_input_is_inflated_member_by_member_in_parallel(tmp_path):
    """BGZF read files (bgzip / htslib / BCL Convert) are gzip members with known sizes: the reader inflates them in parallel
    (the reference's seqan3 bgzf stream does) and must see exactly the records zlib's single stream sees"""
    import gzip
    data = gzip.decompress(open(os.path.join(G, "cfg1_reads.fastq.gz"), "rb").read())
    big = data * 40  # 8000 records, several hundred members at the small block size
    plain = tmp_path / "big.fastq"
    plain.write_bytes(big)
    want = subprocess.run([EXE, "_records", str(plain)], stdout=subprocess.PIPE).stdout
    assert want.count(b"\n") == 8000
    for block, threads, max_bytes in ((65280, 8, 1 << 20), (1000, 3, 4096), (1, 2, 64), (65280, 1, 1 << 26)):
        src = big if block > 1000 else data if block > 1 else b"".join(data.splitlines(keepends=True)[:20])  # one-byte members: five records
        f = tmp_path / ("b%d.fastq.gz" % block)
        f.write_bytes(_bgzf(src, block))
        env = dict(os.environ, CHARON_READER_THREADS=str(threads))
        got = subprocess.run([EXE, "_records", str(f), "1000", str(max_bytes)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        ref = subprocess.run([EXE, "_records", str(f), "1000", str(max_bytes)], stdout=subprocess.PIPE, env=dict(env, CHARON_NO_BGZF="1"))
        assert got.returncode == 0, got.stderr
        assert got.stdout == ref.stdout
        if block > 1000:
            assert got.stdout == want
    # damage: a flipped data byte (CRC32), a cut-off file, a member that is not BGZF in the middle
    good = _bgzf(big, 65280)
    bad = bytearray(good); bad[len(bad) // 2] ^= 0x11
    cut = good[:len(good) * 2 // 3]
    mixed = _bgzf(data, 65280, eof_marker=False) + gzip.compress(data)
    for name, blob in (("crc", bytes(bad)), ("cut", cut), ("mixed", mixed)):
        f = tmp_path / (name + ".fastq.gz")
        f.write_bytes(blob)
        rc, out, err = run(["_records", str(f)])
        assert rc != 0 and "gzip read error" in err, (name, rc, err)


def _fastq_blob(n, seed, lens=(80, 150, 300, 5000)):
    import random
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        L = rnd.choice(lens)
        seq = "".join(rnd.choice("ACGTN" if rnd.random() < 0.02 else "ACGT") for _ in range(L))
        q, cur = [], rnd.choice("#+5?I")
        while sum(map(len, q)) < L:
            q.append(cur * rnd.randint(1, 40))
            cur = rnd.choice("+5?II??5")
        out.append("@read%d runid=%08x ch=%d\n%s\n+\n%s\n" % (i, rnd.getrandbits(32), rnd.randint(1, 512), seq, "".join(q)[:L]))
    return "".join(out).encode()


def _records_of(path, env=None, args=("300", "65536")):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([EXE, "_records", str(path)] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    return p.returncode, p.stdout, p.stderr.decode(errors="replace")


def test_bz2_input_blocks_decoded_side_by_side_equal_libbz2(tmp_path):
    """.bz2 read files (seqan3 reads them through libbz2, which is not in this image): this build's own decoder
    (charon_amd/csrc/host/bz2_stream.inc) finds the blocks by their 48-bit number at any bit position and decodes them in parallel.
    Judge: python's bz2 module (libbz2) -- same bytes for every level, thread count and piece size; same records as the plain file;
    concatenated streams (pbzip2 output), an empty stream; every kind of damage refused, never a short read."""
    import bz2
    blob = _fastq_blob(2500, 21)
    runs = b"A" * 70000 + b"CCCC" + b"G" * 259 + b"T" * 260 + bytes(range(256)) * 30 + b"\x00" * 300000 + b"AAAA"
    cut = blob.rfind(b"\n@read", 0, 1200000) + 1
    cases = {"l9": (blob, bz2.compress(blob, 9)), "l1": (blob, bz2.compress(blob, 1)), "runs": (runs, bz2.compress(runs, 2)),
             "empty": (b"", bz2.compress(b"")), "one": (b"x", bz2.compress(b"x")),
             # periodic text: the inverse-BWT permutation of such a block has several cycles (the decoder's sixteen walkers then give way to
             # the defining walk); one letter only: blocks of a few dozen bytes that expand to 45 MB each
             "period2": (b"AC" * 1200000, bz2.compress(b"AC" * 1200000, 9)), "period28": (b"@r\nACGTACGTAC\n+\nIIIIIIIIII\n" * 70000, bz2.compress(b"@r\nACGTACGTAC\n+\nIIIIIIIIII\n" * 70000, 9)),
             "oneletter": (b"A" * (3 << 24), bz2.compress(b"A" * (3 << 24), 9)),
             "streams": (blob, bz2.compress(blob[:cut], 9) + bz2.compress(b"", 1) + bz2.compress(blob[cut:], 3)),
             # bytes behind a complete stream that do not open another one: libbz2's callers stop there (python's bz2 ignores them, bzip2 warns),
             # and so does this decoder -- as the gzip reader does
             "trailing": (blob, bz2.compress(blob, 9) + b"\0" * 64), "trailing2": (b"x", bz2.compress(b"x") + b"garbage")}
    assert bz2.decompress(cases["trailing"][1]) == blob
    for name, (want, comp) in cases.items():
        f = tmp_path / (name + ".bz2")
        f.write_bytes(comp)
        for threads, piece in (("1", "67108864"), ("6", "1000"), ("3", "67108864")):
            p = subprocess.run([EXE, "_bunzip2", str(f), piece], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, CHARON_READER_THREADS=threads))
            assert p.returncode == 0, (name, p.stderr)
            assert p.stdout == want, (name, threads, piece)
    # a damaged file libbz2 ACCEPTS (tools/fuzz/fuzz_bz2.py, seed 9): one code-length set is not a complete prefix code; libbz2 never checks
    # that, the odd codes do not occur and the CRCs hold -- the decoder's tables must be libbz2's own (limit / base / perm), not a validated code
    f = os.path.join(G, "oversubscribed_codes.bz2")
    want = bz2.decompress(open(f, "rb").read())
    for threads in ("1", "4"):
        p = subprocess.run([EXE, "_bunzip2", f], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, CHARON_READER_THREADS=threads))
        assert p.returncode == 0 and p.stdout == want, p.stderr
    # the records the reader hands on: .fastq.bz2 == the plain file
    plain = tmp_path / "r.fastq"
    plain.write_bytes(blob)
    want = _records_of(plain)
    assert want[0] == 0 and want[1].count(b"\n") == 2500
    for name in ("l9", "l1", "streams"):
        f = tmp_path / (name + ".fastq.bz2")
        f.write_bytes(cases[name][1])
        for threads in ("1", "5"):
            got = _records_of(f, {"CHARON_READER_THREADS": threads})
            assert got[0] == 0, got[2]
            assert got[1] == want[1], (name, threads)
    # damage
    good = cases["l9"][1]
    flipped = bytearray(good); flipped[len(good) // 2] ^= 0x04
    crc = bytearray(good); crc[10] ^= 0x01                      # the first block's CRC field
    tail = bytearray(good); tail[-2] ^= 0x80                     # the stream's combined CRC
    for name, b in (("flipped", bytes(flipped)), ("crc", bytes(crc)), ("tail", bytes(tail)), ("cut", good[:len(good) * 3 // 5]), ("cut2", good[:-3]),
                    ("nothing", b""), ("text", b"@r\nACGT\n+\nIIII\n"), ("header", b"BZh0" + good[4:])):
        f = tmp_path / (name + ".fastq.bz2")
        f.write_bytes(b)
        for threads in ("1", "4"):
            rc, out, err = _records_of(f, {"CHARON_READER_THREADS": threads})
            assert rc == 1 and "bzip2 read error" in err, (name, rc, err)
            assert not out.endswith(b"\n") or out.count(b"\n") < 2500, name


def test_one_stream_gz_decoded_in_chunks_equals_zlib(tmp_path):
    """A one-stream .gz is cut at searched block starts and decoded by several threads with markers for the unknown window
    (charon_amd/csrc/host/inflate_stream.inc); whatever the chunking, the records must be the ones zlib's inflate gives."""
    import gzip, zlib
    blob = _fastq_blob(3000, 11)
    cut = lambda n: blob.rfind(b"\n@read", 0, n) + 1  # a record boundary at or before n
    a, b, c = cut(300000), cut(400000), cut(900000)
    cases = {}
    for lvl in (1, 6, 9):
        cases["l%d" % lvl] = gzip.compress(blob, lvl)
    cases["stored"] = gzip.compress(blob[:a], 0)
    cases["stored_all"] = gzip.compress(blob, 0)   # level 0 throughout: no block start to be found, the searches must stop after a few
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    cases["fixed"] = co.compress(blob[:a]) + co.flush()
    cases["members"] = gzip.compress(blob[:b], 6) + gzip.compress(blob[b:c], 1) + gzip.compress(blob[c:], 9)
    # header with FEXTRA, FNAME, FCOMMENT, FHCRC; trailing bytes that are not a gzip header
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(blob) + raw.flush()
    import struct
    hdr = b"\x1f\x8b\x08\x1e\0\0\0\0\0\x03" + struct.pack("<H", 5) + b"ab\x01\x00z" + b"reads.fastq\0" + b"a comment\0"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    cases["header"] = hdr + body + struct.pack("<II", zlib.crc32(blob) & 0xFFFFFFFF, len(blob) & 0xFFFFFFFF) + b"\0\0\0trailing"
    # full flushes: byte-aligned empty stored blocks between dynamic ones
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for i in range(0, len(blob), 150000):
        parts.append(co.compress(blob[i:i + 150000]) + co.flush(zlib.Z_FULL_FLUSH if (i // 150000) % 2 else zlib.Z_SYNC_FLUSH))
    cases["flushes"] = b"".join(parts) + co.flush()
    # runs and short periods (matches at distance 1 .. 7, also across chunk starts), far more repetitive than the expansion cap allows
    rep = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, (b"A" * 3000 + b"ACG" * 400 + b"AC" * 300)[:4000 + (i % 7)], b"I" * (4000 + (i % 7))) for i in range(400))
    cases["runs"] = gzip.compress(rep, 6)
    for name, data in cases.items():
        f = tmp_path / (name + ".fastq.gz")
        f.write_bytes(data)
        rc0, want, err0 = _records_of(f, {"CHARON_ZLIB_INFLATE": "1"})
        assert rc0 == 0 and want.count(b"\n") > 100, (name, err0)
        for threads, chunk in ((1, 0), (3, 20000), (4, 1024)):
            env = {"CHARON_READER_THREADS": str(threads)}
            if chunk:
                env["CHARON_INFLATE_CHUNK"] = str(chunk)
            rc, got, err = _records_of(f, env)
            assert rc == 0, (name, threads, chunk, err)
            assert got == want, (name, threads, chunk)
    # the parallel path really ran (and counted its chunks) on the level-6 file
    p = subprocess.run([EXE, "_inflate", str(tmp_path / "l6.fastq.gz")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, CHARON_READER_THREADS="4", CHARON_INFLATE_CHUNK="30000"))
    assert p.returncode == 0, p.stderr
    rounds = int(p.stdout.decode().split("parallel:")[1].split()[0])
    assert rounds >= 2, p.stdout


def test_damaged_gz_fails_like_zlib_never_crashes(tmp_path):
    """bit flips and cuts anywhere in a .gz (60 damaged files): this build's decoder (sequential and chunked) must fail where zlib fails and give
    zlib's records where zlib succeeds (a flip in the header's mtime, say) -- and never die on a signal"""
    import gzip, random
    rnd = random.Random(5)
    blob = _fastq_blob(400, 3, lens=(80, 150, 300))
    good = gzip.compress(blob, 6)
    f = tmp_path / "d.fastq.gz"
    for trial in range(60):
        b = bytearray(good)
        kind = trial % 3
        if kind == 0:
            p = rnd.randrange(len(b)); b[p] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            b = b[:rnd.randrange(1, len(b))]
        else:
            p = rnd.randrange(len(b) - 8); b[p:p + rnd.randint(1, 8)] = bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 8)))
        f.write_bytes(bytes(b))
        rc0, want, _ = _records_of(f, {"CHARON_ZLIB_INFLATE": "1"})
        for env in ({"CHARON_READER_THREADS": "1"}, {"CHARON_READER_THREADS": "4", "CHARON_INFLATE_CHUNK": "2048"}):
            rc, got, err = _records_of(f, env)
            assert rc in (0, 1), (trial, rc, err)
            if rc0 == 0:
                assert rc == 0 and got == want, (trial, kind, err)
            else:
                assert rc == 1 and ("gzip read error" in err or "parse error" in err), (trial, kind, err)


def test_row_number_formatting_equals_printf_g():
    """the front end formats a row's eight real numbers itself (six significant digits, as printf's %g does, exactly: ties and anything outside
    1e-4 .. 1e6 fall back to snprintf); its self-test compares millions of values of every kind a row holds with snprintf"""
    for seed in ("1", "99"):
        p = subprocess.run([EXE, "_gfmt", "3000000", seed], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0 and b" 0 mismatches" in p.stdout, p.stdout


def test_mapped_fastq_split_by_several_threads_equals_the_sequential_parse(tmp_path):
    """a mapped FASTQ file is cut at guessed record starts and parsed by several threads; a piece counts only if the piece before it
    came to stand exactly on its start, and anything but a plain four-line record is left to the sequential parser -- so the records,
    and the errors, must be those of one thread, whatever the file looks like"""
    import random
    rnd = random.Random(17)

    def make(n, crlf=False, blanks=0.0, wrapped=0.0, empty=0.0, at_quals=0.5):
        eol = "\r\n" if crlf else "\n"
        out = []
        for i in range(n):
            L = 0 if rnd.random() < empty else rnd.choice([30, 150, 151, 1000, 5000])
            s = "".join(rnd.choice("ACGTN") for _ in range(L))
            q = "".join(chr(rnd.randint(33, 73)) for _ in range(L))
            if L and rnd.random() < at_quals:
                q = "@" + q[1:]
            if L > 200 and rnd.random() < wrapped:
                s = eol.join(s[j:j + 70] for j in range(0, L, 70)); q = eol.join(q[j:j + 70] for j in range(0, L, 70))
            out.append("@r%d desc @x +y%s%s%s+%s%s%s" % (i, eol, s, eol, eol, q, eol))
            if rnd.random() < blanks:
                out.append(eol)
        return "".join(out).encode()

    cases = {"plain": make(6000), "crlf": make(5000, crlf=True), "blanks": make(5000, blanks=0.01), "wrapped": make(5000, wrapped=0.01),
             "empty": make(6000, empty=0.01), "mixed": make(6000, crlf=True, blanks=0.005, wrapped=0.005, empty=0.005)}
    cases["no_final_newline"] = cases["plain"].rstrip(b"\n")
    cases["cut"] = cases["plain"][:len(cases["plain"]) * 3 // 4 - 17]
    dmg = bytearray(cases["plain"]); dmg[len(dmg) // 2] = ord("\n")
    cases["broken_line"] = bytes(dmg)
    for name, data in cases.items():
        f = tmp_path / (name + ".fastq")
        f.write_bytes(data)
        for recs, mbytes in (("100000", str(32 << 20)), ("5000", str(3 << 20))):
            one = subprocess.run([EXE, "_records", str(f), recs, mbytes], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, CHARON_READER_THREADS="1"))
            for t in ("3",) if name in ("crlf", "blanks", "empty") else ("3", "8"):
                many = subprocess.run([EXE, "_records", str(f), recs, mbytes], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, CHARON_READER_THREADS=t))
                assert many.returncode == one.returncode, (name, recs, mbytes, t, many.stderr[-300:], one.stderr[-300:])
                assert many.stdout == one.stdout, (name, recs, mbytes, t)
                assert many.stderr == one.stderr, (name, t, many.stderr[-300:], one.stderr[-300:])


def test_elias_fano_arrays_written_by_several_threads_equal_the_sequential_writer():
    """`charon index` writes the sd_vector's low / high arrays from all its threads at once (a set bit's place depends only on its
    position and on the number of set bits before it; the words at the seams between two threads' stretches are OR-ed atomically):
    the arrays must be the ones the bit-by-bit writer produces, for any density and thread count (charon_amd/csrc/host/index_builder.inc)"""
    for seed, nwords, density, threads in ((1, 5000, 0.2, 7), (2, 1, 0.5, 4), (3, 64, 1.0, 3), (4, 20000, 0.001, 8), (5, 3000, 0.02, 16), (6, 2, 0.0, 5), (7, 777, 0.6, 2)):
        rc, out, err = run(["_efcheck", str(seed), str(nwords), str(density), str(threads)])
        assert rc == 0 and " same 1" in out, (seed, nwords, density, threads, out, err)


def test_short_read_gz_in_small_blocks_keeps_memory_and_time_bounded(tmp_path):
    """on the decoded-slab path (.gz, BGZF, .bz2, gzread) every call decoded another max_bytes whatever the carry already held, so
    with short reads and few records per block the whole inflated file piled up in memory and was copied on every call (16.6 s / 342 MB for
    95 MB of FASTQ).  Now a call decodes what its records need: same records, resident memory a small multiple of a block, linear time."""
    import gzip, time
    import numpy as np
    r = np.random.default_rng(3)
    n = 250000
    seqs = np.frombuffer(b"ACGT", np.uint8)[r.integers(0, 4, (n, 150))]
    blob = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, seqs[i].tobytes(), b"I" * 150) for i in range(n))   # 78 MB
    plain = tmp_path / "short.fastq"
    plain.write_bytes(blob)
    rc, want, err = _records_of(plain, args=("65536", str(1 << 28)))
    assert rc == 0 and want.count(b"\n") == n, err
    files = {"gz": gzip.compress(blob, 1)}
    import bz2
    files["bz2"] = bz2.compress(blob[:blob.find(b"@r%d\n" % (n // 4))], 1)   # the first quarter of the records
    for ext, data in files.items():
        f = tmp_path / ("short.fastq." + ext)
        f.write_bytes(data)
        for env in ({}, {"CHARON_ZLIB_INFLATE": "1"}) if ext == "gz" else ({},):
            e = dict(os.environ)
            e.update(env)
            t0 = time.time()
            p = subprocess.Popen([EXE, "_records", str(f), "1000", str(1 << 24)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
            out, perr = p.communicate()
            dt = time.time() - t0
            assert p.returncode == 0, perr
            assert out == (want if ext == "gz" else want[:len(out)]) and out.count(b"\n") == (n if ext == "gz" else n // 4), (ext, env)
            assert dt < 8.0, (ext, env, dt)   # 16.6 s before for a file of this kind on this container's cores; 0.3 - 0.8 s now
    # resident memory: run once more and watch the child's VmRSS (a forked child's ru_maxrss starts at the python parent's size)
    p = subprocess.Popen([EXE, "_records", str(tmp_path / "short.fastq.gz"), "1000", str(1 << 24)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    peak = 0
    while p.poll() is None:
        try:
            for line in open("/proc/%d/status" % p.pid):
                if line.startswith("VmRSS:"):
                    peak = max(peak, int(line.split()[1]))
        except OSError:
            pass
        time.sleep(0.005)
    assert p.returncode == 0
    assert 0 < peak < 160 * 1024, peak   # KiB; 342 MB before for 95 MB of text, some 75 MB now
