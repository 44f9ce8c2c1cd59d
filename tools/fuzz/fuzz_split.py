#!/usr/bin/env python3
"""Damaged FASTQ through an AddressSanitizer / UBSan build of the reader (`charon _records`): the several-thread splitter of mapped files
must give exactly what one thread gives -- records, exit code, error text -- and never trip a sanitizer.  Build as in fuzz_inflate.py.
(round 2: 60 trials, no difference, no finding)"""
import os, random, subprocess, sys
rnd = random.Random(5)
EXE = "/tmp/asan/charon"
def make(n):
    out = []
    for i in range(n):
        L = rnd.choice([0, 30, 150, 1000, 5000]) if rnd.random() < 0.02 else rnd.choice([150, 1000, 5000])
        s = "".join(rnd.choice("ACGTN") for _ in range(L)); q = "".join(chr(rnd.randint(33, 73)) for _ in range(L))
        if L and rnd.random() < 0.5: q = "@" + q[1:]
        out.append("@r%d x\n%s\n+\n%s\n" % (i, s, q))
    return "".join(out).encode()
base = make(3000)
bad = 0
for trial in range(60):
    b = bytearray(base)
    for _ in range(rnd.choice([0, 1, 3, 10])):
        p = rnd.randrange(len(b)); k = rnd.random()
        if k < 0.4: b[p] = rnd.choice(b"\n\r@+ACGT\x00")
        elif k < 0.7: del b[p:p + rnd.randint(1, 3000)]
        else: b[p:p] = bytes(rnd.choice(b"\n@+A") for _ in range(rnd.randint(1, 50)))
    if rnd.random() < 0.2: b = b[:rnd.randrange(len(b))]
    open("/tmp/asan/s.fastq", "wb").write(bytes(b))
    outs = []
    for t in ("1", "4"):
        p = subprocess.run([EXE, "_records", "/tmp/asan/s.fastq", "100000", str(32 << 20)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", CHARON_READER_THREADS=t))
        e = p.stderr.decode(errors="replace")
        if "Sanitizer" in e or "runtime error" in e or p.returncode not in (0, 1):
            bad += 1; print("TRIAL", trial, t, p.returncode, e[:1000])
        outs.append((p.returncode, p.stdout, e))
    if outs[0] != outs[1]:
        bad += 1; print("DIFF trial", trial, outs[0][0], outs[1][0], outs[0][2][-200:], outs[1][2][-200:])
print("asan split fuzz: 60 trials, %d bad" % bad)
