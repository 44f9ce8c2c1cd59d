"""Time-boxed randomised soak of `charon index` (GPU builder) against the oracle's sequential builder.
    python tools/fuzz/fuzz_index.py <seconds> <seed>     (needs a GPU)
Varies: number of files / categories, records per file, record lengths (incl. shorter than k and than w), wrapped lines, N runs,
lower case, low-complexity stretches, k, w, piece size (CHARON_INDEX_PIECE) and -t; compares metadata and every IBF word."""
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
from oracle import pyoracle as po  # noqa: E402

EXE = os.path.join(ROOT, "charon_amd", "bin", "charon")


def trial(r, d):
    k = int(r.integers(4, 28))
    w = int(k + r.integers(0, 40))
    nfiles = int(r.integers(1, 12))
    catnames = ["human", "microbial", "viral"][:int(r.integers(1, 4))]
    files, cats = [], []
    for i in range(nfiles):
        path = os.path.join(d, "ref%d.fa" % i)
        with open(path, "w") as f:
            for j in range(int(r.integers(1, 5))):
                L = int(r.choice([0, 3, k - 1, k, w, w + 1, 100, 4095, 4096, 4097, 9000, 40000]))
                s = bytearray(util.random_seq(r, L))
                if L > 200 and r.random() < 0.3:
                    p = int(r.integers(0, L - 100))
                    s[p:p + 40] = b"N" * 40
                if L > 200 and r.random() < 0.3:
                    p = int(r.integers(0, L - 150))
                    s[p:p + 120] = b"AT" * 60
                if L > 200 and r.random() < 0.2:
                    s = bytearray(bytes(s).lower())
                s = s.decode()
                f.write(">rec%d_%d\n" % (i, j))
                ww = int(r.choice([60, 70, 1 << 30]))
                f.write("\n".join(s[q:q + ww] for q in range(0, max(len(s), 1), ww)) + "\n")
        files.append(path)
        cats.append(catnames[int(r.integers(0, len(catnames)))])
    tab = os.path.join(d, "in.tab")
    with open(tab, "w") as f:
        for p, c in zip(files, cats):
            f.write("%s\t%s\n" % (p, c))
    env = dict(os.environ)
    if r.random() < 0.5:
        env["CHARON_INDEX_PIECE"] = str(int(r.choice([4096, 8192, 65536])))
    threads = int(r.choice([1, 4]))
    p = subprocess.run([EXE, "index", "-k", str(k), "-w", str(w), "-t", str(threads), "-p", os.path.join(d, "built"), "--log", os.path.join(d, "i.log"), tab],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    if b"bin size of 0" in p.stderr:
        # no record reached k bases: seqan3's IBF constructor throws in the reference; both restatements must refuse too
        assert p.returncode != 0
        h = po.lib().orc_index_build_from_fasta  # noqa: F841  (the python wrapper would wrap a null handle)
        try:
            want = po.Index.from_fasta(list(zip(files, cats)), sorted(set(cats)), k=k, w=w)
            ok = not want.h
        except Exception:
            ok = True
        assert ok, "oracle built an index without minimisers"
        return "k=%d w=%d files=%d: no minimisers, refused by both" % (k, w, nfiles)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = po.Index.load(os.path.join(d, "built.idx"))
    want = po.Index.from_fasta(list(zip(files, cats)), got.categories, k=k, w=w)
    try:
        assert (got.k, got.w, got.bins, got.bin_size, got.hash_funs) == (want.k, want.w, want.bins, want.bin_size, want.hash_funs), \
            ((got.k, got.w, got.bins, got.bin_size, got.hash_funs), (want.k, want.w, want.bins, want.bin_size, want.hash_funs))
        assert list(got.bin_to_cat) == list(want.bin_to_cat)
        assert np.array_equal(got.words(), want.words())
    finally:
        got.free()
        want.free()
    return "k=%d w=%d files=%d cats=%d piece=%s t=%d" % (k, w, nfiles, len(set(cats)), env.get("CHARON_INDEX_PIECE", "-"), threads)


def main():
    secs, seed = float(sys.argv[1]), int(sys.argv[2])
    po.build()
    r = np.random.default_rng(seed)
    t0, t = time.time(), 0
    while time.time() - t0 < secs:
        d = tempfile.mkdtemp(prefix="chfi")
        try:
            msg = trial(r, d)
        except Exception:
            keep = os.path.join(ROOT, "gpurun_out", "fuzz_index_fail_%d_%d" % (seed, t))
            shutil.copytree(d, keep, dirs_exist_ok=True)
            print("FAILURE in trial %d (seed %d); inputs kept in %s" % (t, seed, keep), flush=True)
            raise
        finally:
            shutil.rmtree(d, ignore_errors=True)
        print("trial %d ok  %s  [%.0fs]" % (t, msg, time.time() - t0), flush=True)
        t += 1
    print("fuzz_index: %d trials, no mismatch" % t)


if __name__ == "__main__":
    main()
