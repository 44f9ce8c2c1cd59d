#!/usr/bin/env python3
"""How long does `charon index` take on references of some size?  Two synthetic genomes of N Mb each (one per category).
usage: python tools/index_build_time.py [Mb per genome] [workdir] [threads]"""
import os, subprocess, sys, time
import numpy as np
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/charon_idx"
threads = sys.argv[3] if len(sys.argv) > 3 else "16"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(work, exist_ok=True)
r = np.random.default_rng(3)
acgt = np.frombuffer(b"ACGT", np.uint8)
with open(os.path.join(work, "refs.tsv"), "w") as tab:
    for name in ("microbial", "human"):
        fa = os.path.join(work, name + ".fa")
        with open(fa, "wb") as f:
            for c in range(mb // 50):  # chromosomes of 50 Mb, lines of 60 letters
                s = acgt[r.integers(0, 4, 50_000_000)]
                f.write(b">%s_chr%d\n" % (name.encode(), c))
                f.write(b"\n".join(s[i:i + 60].tobytes() for i in range(0, len(s), 60)) + b"\n")
        tab.write("%s\t%s\n" % (fa, name))
for f in ("big.idx",):
    if os.path.exists(os.path.join(work, f)):
        os.remove(os.path.join(work, f))
exe = os.path.join(ROOT, "charon_amd", "bin", "charon")
t0 = time.time()
p = subprocess.run([exe, "index", "-t", threads, "-p", os.path.join(work, "big"), "--log", os.path.join(work, "i.log"), os.path.join(work, "refs.tsv")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, CHARON_TIMING="1"))
dt = time.time() - t0
print("charon index -t %s: rc=%d, 2 x %d Mb in %.1f s (%.1f Mb/s), index file %.1f MB" % (threads, p.returncode, mb, dt, 2 * mb / dt, os.path.getsize(os.path.join(work, "big.idx")) / 1e6 if p.returncode == 0 else 0))
print(p.stderr.decode()[-600:])
print(open(os.path.join(work, "i.log")).read()[-600:])
