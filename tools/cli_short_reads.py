#!/usr/bin/env python3
"""`charon dehost` on short reads: N single 150-base reads (and N pairs of 2 x 150) -- per-read overheads instead of per-base work.
usage: python tools/cli_short_reads.py [n_reads] [workdir]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/charon_short"
os.makedirs(work, exist_ok=True)
r = util.rng(9)
gs = [util.random_seq(r, 2_000_000), util.random_seq(r, 2_000_000)]
exe = os.path.join(ROOT, "charon_amd", "bin", "charon")
with open(os.path.join(work, "refs.tsv"), "w") as tab:
    for name, g in (("microbial", gs[0]), ("human", gs[1])):
        fa = os.path.join(work, name + ".fa")
        open(fa, "wb").write(b">" + name.encode() + b"\n" + g + b"\n")
        tab.write("%s\t%s\n" % (fa, name))
if os.path.exists(os.path.join(work, "s.idx")):
    os.remove(os.path.join(work, "s.idx"))
subprocess.run([exe, "index", "-p", os.path.join(work, "s"), "--log", os.path.join(work, "i.log"), os.path.join(work, "refs.tsv")], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
starts = r.integers(0, 2_000_000 - 400, n)
q = b"I" * 150
with open(os.path.join(work, "s_1.fastq"), "wb") as f1, open(os.path.join(work, "s_2.fastq"), "wb") as f2:
    for i in range(n):
        g = gs[i & 1]
        s = int(starts[i])
        f1.write(b"@p%d/1\n%s\n+\n%s\n" % (i, g[s:s + 150], q))
        f2.write(b"@p%d/2\n%s\n+\n%s\n" % (i, g[s + 200:s + 350], q))
print("%d reads of 150 bases written" % n, flush=True)
for files, label in (([os.path.join(work, "s_1.fastq")], "single"), ([os.path.join(work, "s_1.fastq"), os.path.join(work, "s_2.fastq")], "paired")):
    for t in (1, 16):
        t0 = time.time()
        p = subprocess.run([exe, "dehost", "--db", os.path.join(work, "s.idx"), "-t", str(t), "--log", os.path.join(work, "c.log")] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, CHARON_TIMING="1"))
        dt = time.time() - t0
        tm = [l for l in p.stderr.decode().split("\n") if "main thread" in l]
        print("%s -t %2d: rc=%d rows=%d  %.2f s -> %.0f reads/s   %s" % (label, t, p.returncode, p.stdout.count(b"\n"), dt, n / dt, tm[0][8:] if tm else p.stderr.decode()[-300:]), flush=True)
