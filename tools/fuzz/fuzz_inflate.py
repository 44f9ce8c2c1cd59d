#!/usr/bin/env python3
"""Damaged .gz files through an AddressSanitizer / UBSan build of the front end's reader (`charon _records`, no GPU involved):
this build's inflate (sequential and chunked) must never trip a sanitizer or die on a signal.
build:  g++ -O1 -g -std=c++14 -fopenmp -fsanitize=address,undefined -I include -o /tmp/asan/charon charon_amd/csrc/host/charon_main.cpp \
            -Lcharon_amd -lcharon_hip -lz -Wl,-rpath,$PWD/charon_amd
usage:  python tools/fuzz/fuzz_inflate.py [seed] [trials]        (results of round 2: seeds 1 and 7, 550 trials x 2 modes, 0 findings)"""
import gzip, os, random, subprocess, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(sys.path[0], "tests", "test_cli_cpu.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
EXE = "/tmp/asan/charon"
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
blob = t._fastq_blob(600, 9, lens=(80, 150, 300, 2000))
variants = [gzip.compress(blob, 6), gzip.compress(blob, 1), gzip.compress(blob[:blob.rfind(b"\n@read", 0, 100000) + 1], 0)]
bad = 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for trial in range(n):
    b = bytearray(rnd.choice(variants))
    for _ in range(rnd.choice([1, 1, 2, 5])):
        k = rnd.random()
        if k < 0.4: p = rnd.randrange(len(b)); b[p] ^= 1 << rnd.randrange(8)
        elif k < 0.6: b = b[:rnd.randrange(1, len(b))]
        elif k < 0.8: p = rnd.randrange(len(b)); b[p:p + rnd.randint(1, 30)] = bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 30)))
        else: p = rnd.randrange(len(b)); del b[p:p + rnd.randint(1, 20)]
        if len(b) < 2: break
    open("/tmp/asan/f.fastq.gz", "wb").write(bytes(b))
    for env in ({"CHARON_READER_THREADS": "1"}, {"CHARON_READER_THREADS": "4", "CHARON_INFLATE_CHUNK": str(rnd.choice([1024, 3000, 20000]))}):
        e = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", **env)
        p = subprocess.run([EXE, "_records", "/tmp/asan/f.fastq.gz", "300", "65536"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        err = p.stderr.decode(errors="replace")
        if "Sanitizer" in err or "runtime error" in err or p.returncode not in (0, 1):
            bad += 1
            print("TRIAL", trial, env, p.returncode, err[:1500])
            os.rename("/tmp/asan/f.fastq.gz", "/tmp/asan/crash_%d.fastq.gz" % trial)
            break
print("asan fuzz: %d trials, %d bad" % (n, bad))
