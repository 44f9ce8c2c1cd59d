#!/usr/bin/env python3
"""Damaged .bz2 files through an AddressSanitizer / UBSan build of the front end's reader (`charon _bunzip2`, no GPU involved):
this build's bzip2 decoder (host/bz2_stream.inc) must never trip a sanitizer or die on a signal, and must accept or refuse a file
exactly as libbz2 does (python's bz2 module is the judge; output compared byte for byte when both accept).
build:  g++ -O1 -g -std=c++14 -fopenmp -fsanitize=address,undefined -I include -o /tmp/asan/charon charon_amd/csrc/host/charon_main.cpp \
            -Lcharon_amd -lcharon_hip -lz -Wl,-rpath,$PWD/charon_amd
usage:  python tools/fuzz/fuzz_bz2.py [seed] [trials]        (results of round 2: see tools/fuzz/README.md)"""
import bz2, os, random, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(sys.path[0], "tests", "test_cli_cpu.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
EXE = os.environ.get("CHARON_FUZZ_EXE", "/tmp/asan/charon")
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
blob = t._fastq_blob(600, 9, lens=(80, 150, 300, 2000))
runs = b"A" * 5000 + b"ACGT" * 3000 + bytes(range(256)) * 20 + b"\n" * 300
variants = [bz2.compress(b"ACGTTGA" * 40000, 1), bz2.compress(blob, 9), bz2.compress(blob, 1), bz2.compress(runs, 1), bz2.compress(blob[:70000], 1) + bz2.compress(b"", 9) + bz2.compress(blob[70000:], 2)]
bad = 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
accepted = 0
for trial in range(n):
    b = bytearray(rnd.choice(variants))
    for _ in range(rnd.choice([0, 1, 1, 2, 5])):
        k = rnd.random()
        if k < 0.4: p = rnd.randrange(len(b)); b[p] ^= 1 << rnd.randrange(8)
        elif k < 0.6: b = b[:rnd.randrange(1, len(b))]
        elif k < 0.8: p = rnd.randrange(len(b)); b[p:p + rnd.randint(1, 30)] = bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 30)))
        else: p = rnd.randrange(len(b)); del b[p:p + rnd.randint(1, 20)]
        if len(b) < 2: break
    open("/tmp/asan/f.bz2", "wb").write(bytes(b))
    try:
        # python's bz2.decompress ignores garbage behind a whole stream; this build refuses it: judge with a strict loop
        want, rest, ok = b"", bytes(b), True
        while rest:
            d = bz2.BZ2Decompressor()
            want += d.decompress(rest)
            if not d.eof: ok = False; break
            rest = d.unused_data
        if not bytes(b): ok = False
    except (OSError, ValueError):
        ok = False
    for env in ({"CHARON_READER_THREADS": "1"}, {"CHARON_READER_THREADS": "4"}):
        e = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", **env)
        p = subprocess.run([EXE, "_bunzip2", "/tmp/asan/f.bz2", str(rnd.choice([1000, 1 << 26]))], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        err = p.stderr.decode(errors="replace")
        wrong = (p.returncode == 0) != ok or (ok and p.stdout != want)
        if "Sanitizer" in err or "runtime error" in err or p.returncode not in (0, 1) or wrong:
            bad += 1
            print("TRIAL", trial, env, p.returncode, "libbz2 accepts" if ok else "libbz2 refuses", err[:1500])
            os.rename("/tmp/asan/f.bz2", "/tmp/asan/crash_%d.bz2" % trial)
            break
    accepted += ok
print("bz2 fuzz: %d trials (%d accepted by both), %d bad" % (n, accepted, bad))
