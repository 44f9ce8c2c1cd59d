"""Time-boxed randomised soak of the drop-in boundary: `charon dehost` (GPU) vs the oracle's dehost on random indexes and read files.
    python tools/fuzz/fuzz_cli.py <seconds> <seed>     (needs a GPU)
Varies: bins / categories / k / w, FASTQ vs FASTA, gz / BGZF / bz2, wrapped lines, CRLF, lower case + IUPAC, zero-length reads, single vs paired,
batch size (CHARON_BATCH_READS), -t, --extract with a small --num_reads_to_fit (training path), thresholds."""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util  # noqa: E402
from tests.test_gpu_cli import assert_same_tsv  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

EXE = os.path.join(ROOT, "charon_amd", "bin", "charon")


def write_reads(path, names, seqs, r, fastq, gz, wrapw, crlf):
    eol = "\r\n" if crlf else "\n"
    wrap = (lambda s: eol.join(s[i:i + wrapw] for i in range(0, len(s), wrapw)) if s else "") if wrapw else (lambda s: s)
    out = []
    for nm, s in zip(names, seqs):
        s = s.decode()
        if fastq:
            q = "".join(chr(33 + int(x)) for x in r.integers(2, 41, len(s)))
            out.append("@" + nm + eol + wrap(s) + eol + "+" + eol + wrap(q) + eol)
        else:
            out.append(">" + nm + eol + wrap(s) + eol)
    data = "".join(out).encode()
    if gz:
        kind = r.random()
        if kind < 0.35:  # BGZF: members of random size (the reader inflates them in parallel)
            import struct, zlib
            block = int(r.choice([200, 3000, 65280]))
            with open(path, "wb") as f:
                for c in [data[i:i + block] for i in range(0, len(data), block)] + [b""]:
                    co = zlib.compressobj(int(r.choice([1, 6])), zlib.DEFLATED, -15)
                    z = co.compress(c) + co.flush()
                    f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(z) + 8 - 1) + z +
                            struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))
        else:  # one stream (or two members); the chunked inflate is forced onto small files by CHARON_INFLATE_CHUNK in trial()
            with open(path, "wb") as f:
                cut = len(data) // 2 if kind > 0.8 else len(data)
                cut = data.rfind(b"\n", 0, cut) + 1 if cut < len(data) else cut
                f.write(gzip.compress(data[:cut], int(r.choice([1, 6, 9]))))
                if cut < len(data):
                    f.write(gzip.compress(data[cut:], 6))
    else:
        with open(path, "wb") as f:
            f.write(data)
    return data


def trial(r, d):
    k = int(r.integers(5, 28))
    w = int(k + r.integers(0, 30))
    nfiles = int(r.choice([2, 3, 5, 9]))
    cats = ["human", "bacteria", "virus"][:int(min(nfiles, r.integers(2, 4)))]
    paired = bool(r.integers(0, 2)) or len(cats) > 2
    gs = [util.random_seq(r, int(r.integers(2000, 9000))) for _ in range(nfiles)]
    files = []
    for i, g in enumerate(gs):
        p = os.path.join(d, "g%d.fa" % i)
        with open(p, "w") as f:
            f.write(">g%d\n%s\n" % (i, g.decode()))
        files.append((p, cats[i % len(cats)]))
    oidx = po.Index.from_fasta(files, sorted(cats), k=k, w=w)
    oidx.store(os.path.join(d, "x.idx"))
    n = int(r.choice([1, 2, 30, 200, 700]))
    lmax = int(r.choice([60, 300, 2000]))
    fastq = bool(r.random() < 0.8)
    gz = bool(r.random() < 0.3)
    wrapw = int(r.choice([0, 0, 61]))
    crlf = bool(r.random() < 0.15)
    bz = bool(not gz and r.random() < 0.15)  # .bz2 for the front end (its own decoder), the plain twin for the oracle's reader
    ext = (".fastq" if fastq else ".fasta") + (".gz" if gz else "")

    def spice(s):
        a = bytearray(s)
        if len(a) > 30 and r.random() < 0.2:
            a[3:9] = b"nryKMs"
        if len(a) > 30 and r.random() < 0.1:
            a = bytearray(bytes(a).lower())
        if r.random() < 0.03:
            a = bytearray()
        return bytes(a)
    m1 = [spice(s) for s in util.sample_reads(r, gs, n, (20, lmax), sub_rate=0.03, random_fraction=0.15)]
    f1 = os.path.join(d, "r_1" + ext)
    args, kw = [], {}
    if paired:
        m2 = [spice(s) for s in util.sample_reads(r, gs, n, (20, lmax), sub_rate=0.03)]
        f2 = os.path.join(d, "r_2" + ext)
        write_reads(f1, ["q%d extra/1" % i for i in range(n)], m1, r, fastq, gz, wrapw, crlf)
        write_reads(f2, ["q%d extra/2" % i for i in range(n)], m2, r, fastq, gz, wrapw, crlf)
        files_arg = [f1, f2]
    else:
        write_reads(f1, ["q%d some text" % i for i in range(n)], m1, r, fastq, gz, wrapw, crlf)
        files_arg = [f1]
    if r.random() < 0.5:
        mq = float(r.choice([0.0, 10.0, 25.0]))
        args += ["--min_quality", str(mq)]
        kw["min_quality"] = mq
    if r.random() < 0.3:
        cf = int(r.choice([0, 3, 30, 130, 200]))
        args += ["--confidence", str(cf)]
        kw["confidence"] = cf
    if bz:
        import bz2
        twins = []
        for f in files_arg:
            data = open(f, "rb").read()
            cut = data.rfind(b"\n", 0, len(data) // 2) + 1
            blob = bz2.compress(data, int(r.choice([1, 9]))) if r.random() < 0.6 else bz2.compress(data[:cut], 1) + bz2.compress(data[cut:], 5)
            open(f + ".bz2", "wb").write(blob)
            twins.append(f + ".bz2")
        oracle_files, files_arg = files_arg, twins
    else:
        oracle_files = files_arg
    extract = r.random() < 0.25 and not bz  # (get_extension gives ".bz2": the reference's writer refuses such a name, and so does this build)
    if extract:
        nfit = int(r.choice([5, 20, 50]))
        args += ["--extract", cats[0], "--num_reads_to_fit", str(nfit), "--prefix", os.path.join(d, "ex")]
        kw.update(run_extract=True, num_reads_to_fit=nfit)
    want = oidx.dehost_files(oracle_files[0], oracle_files[1] if paired else "", **kw)
    env = dict(os.environ)
    env["CHARON_BATCH_READS"] = str(int(r.choice([1, 7, 64, 1000, 65536])))
    if r.random() < 0.7:
        env["CHARON_INFLATE_CHUNK"] = str(int(r.choice([1024, 4096, 30000])))
    threads = int(r.choice([1, 3, 8]))
    p = subprocess.run([EXE, "dehost", "--db", os.path.join(d, "x.idx"), "-t", str(threads), "--log", os.path.join(d, "log")] + args + files_arg,
                       cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert_same_tsv(p.stdout.decode() if p.stdout else "", want) if (want.strip() or p.stdout.strip()) else None
    oidx.free()
    return "k=%d w=%d files=%d cats=%d n=%d lmax=%d paired=%d fastq=%d gz=%d bz2=%d wrap=%d crlf=%d extract=%d batch=%s t=%d rows=%d" % (
        k, w, nfiles, len(cats), n, lmax, paired, fastq, gz, bz, wrapw, crlf, extract, env["CHARON_BATCH_READS"], threads, len(want.strip().split("\n")) if want.strip() else 0)


def main():
    secs, seed = float(sys.argv[1]), int(sys.argv[2])
    po.build()
    r = np.random.default_rng(seed)
    t0, t = time.time(), 0
    while time.time() - t0 < secs:
        d = tempfile.mkdtemp(prefix="chfz")
        try:
            msg = trial(r, d)
        except Exception:
            keep = os.path.join(ROOT, "gpurun_out", "fuzz_cli_fail_%d_%d" % (seed, t))
            shutil.copytree(d, keep, dirs_exist_ok=True)
            print("FAILURE in trial %d (seed %d); inputs kept in %s" % (t, seed, keep), flush=True)
            raise
        finally:
            shutil.rmtree(d, ignore_errors=True)
        print("trial %d ok  %s  [%.0fs]" % (t, msg, time.time() - t0), flush=True)
        t += 1
    print("fuzz_cli: %d trials, no mismatch" % t)


if __name__ == "__main__":
    main()
