#!/usr/bin/env python3
"""End-to-end throughput of the drop-in CLI (charon_amd/bin/charon dehost) on a synthetic FASTQ file.

Writes two 2 Mb synthetic genomes, builds a 2-category index from them with this build's own `charon index`, writes N synthetic
5 kb reads (mutated stretches of the genomes) as FASTQ, then times `charon dehost` at several -t values.
usage: python tools/cli_throughput.py [n_reads] [workdir] [--gen-only]      (--gen-only: write bench.idx / reads.fastq and stop)
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util  # noqa: E402  (random_seq / mutate helpers only)


def main():
    gen_only = "--gen-only" in sys.argv
    argv = [a for a in sys.argv if a != "--gen-only"]
    n = int(argv[1]) if len(argv) > 1 else 50000
    work = argv[2] if len(argv) > 2 else "/tmp/charon_cli_bench"
    os.makedirs(work, exist_ok=True)
    r = util.rng(1)
    gs = [util.random_seq(r, 2_000_000), util.random_seq(r, 2_000_000)]
    t0 = time.time()
    exe = os.path.join(ROOT, "charon_amd", "bin", "charon")
    with open(os.path.join(work, "refs.tsv"), "w") as tab:
        for name, g in (("microbial", gs[0]), ("human", gs[1])):
            fa = os.path.join(work, name + ".fa")
            with open(fa, "wb") as f:
                f.write(b">" + name.encode() + b"\n" + g + b"\n")
            tab.write("%s\t%s\n" % (fa, name))
    for f in ("bench.idx",):
        if os.path.exists(os.path.join(work, f)):
            os.remove(os.path.join(work, f))
    p = subprocess.run([exe, "index", "-p", os.path.join(work, "bench"), "--log", os.path.join(work, "i.log"), os.path.join(work, "refs.tsv")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if p.returncode or not os.path.exists(os.path.join(work, "bench.idx")):
        sys.exit("charon index failed: " + p.stderr.decode()[-500:])
    print("index: built by `charon index` in %.1fs" % (time.time() - t0), flush=True)
    fq = os.path.join(work, "reads.fastq")
    t0 = time.time()
    with open(fq, "wb") as f:
        qual = b"I" * 5000
        for i in range(n):
            g = gs[i & 1]
            s = int(r.integers(0, len(g) - 5000))
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, util.mutate(r, g[s:s + 5000], 0.05), qual))
    print("fastq: %d reads, %.2f GB, written in %.1fs" % (n, os.path.getsize(fq) / 1e9, time.time() - t0), flush=True)
    if gen_only:
        return
    ref = None
    # (the "no gzip column" leg of round 2 needed CHARON_SKIP_COMPRESSION, which only a -DCHARON_DIAG build of the front end still honours)
    for label, extra in (("gzip column: deflate tallies on the GPU (default)", {}), ("gzip column by zlib (CHARON_ZLIB_ONLY=1)", {"CHARON_ZLIB_ONLY": "1"})):
        print(label, flush=True)
        for t in (1, 16, 64):
            t0 = time.time()
            p = subprocess.run([exe, "dehost", "--db", os.path.join(work, "bench.idx"), "-t", str(t), "--log", os.path.join(work, "c.log"), fq],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **extra))
            dt = time.time() - t0
            rows = p.stdout.count(b"\n")
            cls = sum(1 for line in p.stdout.split(b"\n") if line.startswith(b"C\t"))
            same = ""
            if "SKIP" not in "".join(extra):
                if ref is None:
                    ref = p.stdout
                same = "  identical TSV: %s" % (p.stdout == ref)
            print("  charon dehost -t %d: rc=%d rows=%d classified=%d  %.2fs  -> %.0f reads/s%s" % (t, p.returncode, rows, cls, dt, n / dt, same), flush=True)
            if p.returncode:
                print(p.stderr.decode()[-500:])


if __name__ == "__main__":
    main()
