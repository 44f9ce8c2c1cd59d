#!/usr/bin/env python3
"""How long does `charon dehost` take to get a LARGE index file into HBM?  (load_index, src/load_index.cpp:8-15 / include/index.hpp:122-138: the
reference reads the file and keeps the Elias-Fano form; this build also decodes it into plain rows on the device.)

The file is written by the test oracle (tools may use it; the product never does) from random rows at Charon's fill: `bins` user bins of
2^log2_rows rows each, every user bin set with probability 0.215 per row.  A handful of reads are then classified with CHARON_TIMING=1 and
the start-up phases printed.       usage: python tools/index_load_time.py [bins] [log2_rows] [workdir]
                                          python tools/index_load_time.py 39g 0 0 [workdir]   (bench.py's 39 GB stand-in: filled on the device, fetched, written)
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402

from_device = len(sys.argv) > 1 and sys.argv[1] == "39g"   # the 39 GB stand-in of bench.py, filled on the device and fetched from there
if from_device:
    sys.argv.pop(1)
    import charon_amd.api as api  # noqa: E402
bins = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 28
work = sys.argv[3] if len(sys.argv) > 3 else "/tmp/charon_load"
os.makedirs(work, exist_ok=True)
S = 2437500000 if from_device else 1 << lg
if from_device:
    bins = 100
po.build()
cats = ["host", "microbial"]
oidx = po.Index.new(bins, S, [b % 2 for b in range(bins)], cats)
w = oidx.words()
W = oidx.bin_words
t0 = time.time()
r = np.random.default_rng(5)
if from_device:
    g = api.Index(api.make_desc(bins, S, [b % 2 for b in range(bins)], 2, 0))
    g.synth_fill(43, 0.215)
    chunk_rows = (1 << 30) // (8 * W)
    for r0 in range(0, S, chunk_rows):
        nr = min(chunk_rows, S - r0)
        if api.lib().chn_index_download_rows(g.h, r0, nr, w[r0 * W:].ctypes.data) != 0:
            raise RuntimeError("index download failed")
    g.destroy()
else:
    CH = 1 << 22
    for a in range(0, S, CH):  # rows of W words; user bin b is bit b % 64 of word b // 64
        m = min(CH, S - a)
        blk = np.zeros((m, W), np.uint64)
        for b in range(bins):
            blk[:, b // 64] |= (r.random(m) < 0.215).astype(np.uint64) << np.uint64(b % 64)
        w[a * W:(a + m) * W] = blk.reshape(-1)
print("rows filled in %.1f s: %d bins x %d rows, plain %.2f GB" % (time.time() - t0, bins, S, S * W * 8 / 1e9), flush=True)
t0 = time.time()
oidx.compress()
path = os.path.join(work, "big.idx")
oidx.store(path)
ones = oidx.ef_ones
oidx.free()
print("Elias-Fano form written in %.1f s: %.2f GB, %.3g ones" % (time.time() - t0, os.path.getsize(path) / 1e9, ones), flush=True)
fq = os.path.join(work, "few.fastq")
with open(fq, "w") as f:
    for i in range(64):
        s = "".join("ACGT"[x] for x in r.integers(0, 4, 1000))
        f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * 1000))
exe = os.path.join(ROOT, "charon_amd", "bin", "charon")
for rep in range(2):
    t0 = time.time()
    p = subprocess.run([exe, "dehost", "--db", path, "-t", "16", "--log", os.path.join(work, "c.log"), fq], env=dict(os.environ, CHARON_TIMING="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.time() - t0
    print("run %d: rc=%d wall %.2f s  (%.2f GB of file, %.2f GB plain: %.2f GB/s of plain rows)" % (rep, p.returncode, dt, os.path.getsize(path) / 1e9, S * W * 8 / 1e9,
                                                                                          S * W * 8 / 1e9 / dt))
    for line in p.stderr.decode().splitlines():
        if "start-up" in line or "select" in line or "index decode" in line:
            print("   " + line.strip())
