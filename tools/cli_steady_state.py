#!/usr/bin/env python3
"""Steady-state throughput of the drop-in CLI: `charon dehost` on a FASTQ large enough that start-up and exit are a few per cent of the wall
time (>= 4 M reads of 5 kb, 40 GB of text).

The reads are written by numpy in blocks (fixed-width ids, so a block is one 2-D byte array): mutated stretches of two 2 Mb genomes, as
tools/cli_throughput.py writes them one by one.  Index by this build's own `charon index`.
usage: python tools/cli_steady_state.py [n_reads] [workdir] [threads ...]
Prints wall time and reads/s per -t, the CLI's own phase timers (CHARON_TIMING), and checks that every -t writes the same TSV (sha256)."""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = 5000


IDW = 9  # "@r" + 9 digits
REC = 2 + IDW + 1 + L + 3 + L + 1
_G = None


def _block(args):
    """one block of records, written at its place in the file (records are all REC bytes long)"""
    path, b0, m, seed = args
    r = np.random.default_rng([seed, b0])
    acgt = np.frombuffer(b"ACGT", np.uint8)
    a = np.empty((m, REC), np.uint8)
    a[:, 0] = ord("@"); a[:, 1] = ord("r")
    ids = np.arange(b0, b0 + m)
    for d in range(IDW):
        a[:, 2 + IDW - 1 - d] = ord("0") + (ids // 10 ** d) % 10
    a[:, 2 + IDW] = 10
    s0 = 3 + IDW
    starts = r.integers(0, len(_G[0]) - L, m)
    for i in range(m):
        st = int(starts[i])
        a[i, s0:s0 + L] = _G[(b0 + i) & 1][st:st + L]
    hit = r.integers(0, 256, (m, L), dtype=np.uint8) < 13           # 5 % substitutions (to any letter, the same one included)
    alt = acgt[r.integers(0, 4, (m, L), dtype=np.uint8)]
    seq = a[:, s0:s0 + L]
    seq[hit] = alt[hit]
    a[:, s0 + L] = 10; a[:, s0 + L + 1] = ord("+"); a[:, s0 + L + 2] = 10
    a[:, s0 + L + 3:s0 + 2 * L + 3] = ord("I")
    a[:, REC - 1] = 10
    fd = os.open(path, os.O_WRONLY)
    os.pwrite(fd, a.tobytes(), b0 * REC)
    os.close(fd)
    return m


def write_fastq(path, n, genomes, seed=1, block=10000, workers=None):
    import multiprocessing as mp
    global _G
    _G = [np.frombuffer(x, np.uint8) for x in genomes]
    with open(path, "wb") as f:
        f.truncate(n * REC)
    jobs = [(path, b0, min(block, n - b0), seed) for b0 in range(0, n, block)]
    workers = workers or max(1, min(16, len(os.sched_getaffinity(0))))
    with mp.get_context("fork").Pool(workers) as pool:
        assert sum(pool.imap_unordered(_block, jobs)) == n


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
    work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/charon_cli_steady"
    threads = [int(x) for x in sys.argv[3:]] or [16, 16, 1]
    os.makedirs(work, exist_ok=True)
    st = os.statvfs(work)
    free = st.f_bavail * st.f_frsize
    if n * REC * 1.3 > free:  # the TSV of a run needs room too
        n2 = int(free / 1.3 / REC)
        print("only %.0f GB free under %s: %d reads instead of %d" % (free / 1e9, work, n2, n), flush=True)
        n = n2
    sys.path.insert(0, ROOT)
    from tests import util
    r = util.rng(1)
    gs = [util.random_seq(r, 2_000_000), util.random_seq(r, 2_000_000)]
    exe = os.path.join(ROOT, "charon_amd", "bin", "charon")
    with open(os.path.join(work, "refs.tsv"), "w") as tab:
        for name, g in (("microbial", gs[0]), ("human", gs[1])):
            fa = os.path.join(work, name + ".fa")
            with open(fa, "wb") as f:
                f.write(b">" + name.encode() + b"\n" + g + b"\n")
            tab.write("%s\t%s\n" % (fa, name))
    if os.path.exists(os.path.join(work, "bench.idx")):
        os.remove(os.path.join(work, "bench.idx"))
    p = subprocess.run([exe, "index", "-p", os.path.join(work, "bench"), "--log", os.path.join(work, "i.log"), os.path.join(work, "refs.tsv")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if p.returncode:
        sys.exit("charon index failed: " + p.stderr.decode()[-500:])
    fq = os.path.join(work, "reads.fastq")
    t0 = time.time()
    write_fastq(fq, n, gs)
    print("fastq: %d reads of %d bases, %.1f GB, written in %.0f s" % (n, L, os.path.getsize(fq) / 1e9, time.time() - t0), flush=True)
    digests = set()
    for t in threads:
        out = os.path.join(work, "out_t%d.tsv" % t)
        t0 = time.time()
        with open(out, "wb") as fo:
            p = subprocess.run([exe, "dehost", "--db", os.path.join(work, "bench.idx"), "-t", str(t), "--log", os.path.join(work, "c.log"), fq],
                               stdout=fo, stderr=subprocess.PIPE, env=dict(os.environ, CHARON_TIMING="1"))
        dt = time.time() - t0
        h = hashlib.sha256()
        rows = 0
        with open(out, "rb") as fi:
            for chunk in iter(lambda: fi.read(1 << 24), b""):
                h.update(chunk)
                rows += chunk.count(b"\n")
        digests.add(h.hexdigest())
        print("charon dehost -t %d: rc=%d rows=%d wall %.2f s -> %.0f reads/s   tsv sha256 %s" % (t, p.returncode, rows, dt, n / dt, h.hexdigest()[:16]), flush=True)
        for line in p.stderr.decode().splitlines():
            if "timing" in line:
                print("   " + line.strip(), flush=True)
        os.remove(out)
    print("TSV identical across runs: %s" % (len(digests) == 1))


if __name__ == "__main__":
    main()
