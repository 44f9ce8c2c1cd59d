#!/usr/bin/env python3
"""`charon dehost` on long reads (lengths log-uniform 1 kb .. 60 kb, nanopore-like): where should the deflate pass of reads beyond 16 384
letters run -- on the device (one or two wavefronts per CU) or on the host (size emulator in the packing loop)?  Times -t 1 and -t 16 with
the device limit at 16 384 and at 61 440 letters.   usage: python tools/cli_long_reads.py [n_reads] [workdir] [n_ultra]
n_ultra (round 3): that many of the reads are ultra-long instead (log-uniform 100 kb .. 2 Mb) -- each is rolled by the 64 lanes of one wavefront
(k_minimise_probe's SPLIT launch); before, each held its launch for 0.15-0.2 us per base (a 2 Mb read: 0.4 s per batch it sat in)."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/charon_long"
os.makedirs(work, exist_ok=True)
r = util.rng(7)
gs = [util.random_seq(r, 4_000_000), util.random_seq(r, 4_000_000)]
exe = os.path.join(ROOT, "charon_amd", "bin", "charon")
with open(os.path.join(work, "refs.tsv"), "w") as tab:
    for name, g in (("microbial", gs[0]), ("human", gs[1])):
        fa = os.path.join(work, name + ".fa")
        open(fa, "wb").write(b">" + name.encode() + b"\n" + g + b"\n")
        tab.write("%s\t%s\n" % (fa, name))
if os.path.exists(os.path.join(work, "long.idx")):
    os.remove(os.path.join(work, "long.idx"))
subprocess.run([exe, "index", "-p", os.path.join(work, "long"), "--log", os.path.join(work, "i.log"), os.path.join(work, "refs.tsv")], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
lens = np.exp(r.uniform(np.log(1000), np.log(60000), n)).astype(int)
n_ultra = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if n_ultra:
    lens[r.choice(n, n_ultra, replace=False)] = np.exp(r.uniform(np.log(100000), np.log(2000000), n_ultra)).astype(int)
fq = os.path.join(work, "long.fastq")
with open(fq, "wb") as f:
    for i, L in enumerate(lens):
        g = gs[i & 1]
        s = int(r.integers(0, len(g) - L))
        f.write(b"@r%d\n%s\n+\n%s\n" % (i, util.mutate(r, g[s:s + L], 0.05), b"I" * L))
print("%d reads, %.2f G bases, %.0f %% of the bases in reads beyond 16 384 letters; %d reads beyond 100 kb (longest %d)" %
      (n, lens.sum() / 1e9, 100.0 * lens[lens > 16384].sum() / lens.sum(), int((lens > 100000).sum()), int(lens.max())), flush=True)
ref = None
for t in (1, 16):
    for lim in (16384, 61440):
        t0 = time.time()
        p = subprocess.run([exe, "dehost", "--db", os.path.join(work, "long.idx"), "-t", str(t), "--log", os.path.join(work, "c.log"), fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, CHARON_GZIP_GPU_MAX=str(lim), CHARON_TIMING="1"))
        dt = time.time() - t0
        ref = ref or p.stdout
        tm = [l for l in p.stderr.decode().split("\n") if "main thread" in l]
        print("-t %2d, device limit %5d: %.2f s -> %.0f reads/s, %.0f M bases/s  identical TSV: %s   %s" % (t, lim, dt, n / dt, lens.sum() / dt / 1e6, p.stdout == ref, tm[0] if tm else ""), flush=True)
