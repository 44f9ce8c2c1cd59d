#!/usr/bin/env python3
"""GPU soak of the deflate pass (k_gzip_tally + k_gzip_size) against zlib itself: batches of reads of every make -- random, skewed
compositions, repeats of every period with scattered edits, long matches, N runs, reads that end in a match, pairs -- through
chn_batch_submit with gzip_output = sizes; every size the device reports must equal len(zlib level-6 gzip member) (what gzip-hpp writes:
src/utils.cpp:114-124).  Reads the device hands back (status != 0) must be the ones it may hand back (more than one deflate block).

    python tools/fuzz/fuzz_gzip.py <seconds> <seed>
"""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import charon_amd.api as api  # noqa: E402
from charon_amd import pack  # noqa: E402


def zsize(b):
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8)
    return len(co.compress(b) + co.flush())


def rnd(r, n, p=None, letters=b"ACGT"):
    return bytes(r.choice(list(letters), n, p=p).astype(np.uint8))


def make_read(r):
    kind = int(r.integers(0, 9))
    n = int(2 ** r.uniform(0, 15.9))
    if r.random() < 0.05:
        n = int(r.integers(30000, 61441))
    n = max(1, min(n, 61440))
    if kind == 0:
        return rnd(r, n)
    if kind == 1:
        return rnd(r, n, r.dirichlet([0.4] * 4))
    if kind == 2:  # a unit repeated, with edits
        unit = rnd(r, int(r.integers(1, 300)), r.dirichlet([0.7] * 4))
        s = bytearray((unit * (n // len(unit) + 1))[:n])
        for _ in range(int(r.integers(0, 1 + n // 20))):
            s[int(r.integers(0, n))] = b"ACGT"[int(r.integers(0, 4))]
        return bytes(s)
    if kind == 3:  # copies of earlier stretches at all distances (long matches, far matches, TOO_FAR)
        s = bytearray(rnd(r, min(n, int(r.integers(4, 400)))))
        while len(s) < n:
            if r.random() < 0.6 and len(s) > 8:
                a = int(r.integers(0, len(s) - 3))
                ln = int(min(r.integers(3, 400), len(s) - a))
                s += s[a:a + ln]
            else:
                s += rnd(r, int(r.integers(1, 60)))
        return bytes(s[:n])
    if kind == 4:  # N runs
        s = bytearray(rnd(r, n))
        for _ in range(int(r.integers(1, 6))):
            a, ln = int(r.integers(0, n)), int(r.integers(1, 200))
            s[a:a + ln] = b"N" * len(s[a:a + ln])
        return bytes(s[:n])
    if kind == 5:
        return rnd(r, n, [0.3, 0.2, 0.2, 0.2, 0.1], b"ACGTN")
    if kind == 6:  # homopolymers and short periods
        s = bytearray()
        while len(s) < n:
            s += rnd(r, int(r.integers(1, 4))) * int(r.integers(1, 400))
        return bytes(s[:n])
    if kind == 7:  # the end of the read inside a match, or one or two letters after one
        s = bytearray(make_read_simple(r, n))
        k = int(r.integers(3, 40))
        if len(s) > 2 * k:
            s[-k:] = s[:k]
            s += rnd(r, int(r.integers(0, 3)))
        return bytes(s[:61440])
    return rnd(r, n, r.dirichlet([0.15] * 4))


def make_read_simple(r, n):
    return rnd(r, n)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    r = np.random.default_rng(seed)
    g = api.Index(api.make_desc(2, 1 << 16, [0, 1], 2, 0))
    g.synth_fill(43, 0.05)
    t0 = time.time()
    trials = reads_checked = handed_back = 0
    while time.time() - t0 < seconds:
        paired = r.random() < 0.25
        nreads = int(r.integers(1, 400))
        reads = [make_read(r) for _ in range(nreads)]
        mates = None
        if paired:
            reads = [x[:30000] for x in reads]
            mates = [make_read(r)[:30000] for _ in range(nreads)]
        if r.random() < 0.5:  # a batch without N: the kernel's two-bit form
            sub = bytes(r.choice(list(b"ACGT"), 1).astype(np.uint8))
            reads = [x.replace(b"N", sub) for x in reads]
            mates = [x.replace(b"N", sub) for x in mates] if paired else None
        p = pack.pack_reads(reads, mates) if paired else pack.pack_reads(reads)
        st = api.Stream(g, nreads, p["n_bases"])
        st.set_model(api.default_model(2, 0, paired=paired) if paired else api.default_model(2, 0))
        st.submit_host(p, np.full(nreads, 40.0, np.float32), None, gzip_tallies=61440, gzip_output=2)
        out = st.wait_host()
        st.destroy()
        sizes, status = out["gzip_sizes"], out["gzip_tallies"][:, 316]
        for i in range(nreads):
            whole = reads[i] + (mates[i] if paired else b"")
            if status[i] != 0:
                handed_back += 1
                assert len(whole) > 61440 or len(whole) > 16383 or len(whole) == 0, ("handed back", seed, trials, i, len(whole))
                continue
            want = zsize(whole)
            if int(sizes[i]) != want:
                print("MISMATCH seed %d trial %d read %d len %d: device %d zlib %d  %r" % (seed, trials, i, len(whole), int(sizes[i]), want, whole[:60]))
                sys.exit(1)
            reads_checked += 1
        trials += 1
    g.destroy()
    print("fuzz_gzip seed %d: %d batches, %d reads equal to zlib, %d handed to the host, %.0f s" % (seed, trials, reads_checked, handed_back, time.time() - t0))


if __name__ == "__main__":
    main()
