"""Time-boxed randomised parity soak: GPU path (through the C ABI) vs the CPU oracle on random index shapes and reads.
    python tools/fuzz/fuzz_parity.py <seconds> <seed>     (needs a GPU; prints one line per trial, exits 1 on the first mismatch)
Wider than tests/test_gpu_parity.py::test_randomised_parameter_sweep: hash function counts 1..5, up to 255 bins, reads up to 30 kb,
thousands of reads per batch, batch capacities close to the input size."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
import charon_amd.api as api  # noqa: E402
from charon_amd import pack  # noqa: E402


def trial(r, t):
    k = int(r.integers(3, 28))
    w = int(k + r.integers(0, 40))
    B = int(r.choice([2, 3, 5, 8, 9, 33, 64, 65, 100, 129, 200, 255]))
    C = int(min(B, r.integers(2, 9)))
    h = int(r.integers(1, 6))
    b2c = [int(x) for x in r.integers(0, C, B)]
    for c in range(C):
        b2c[c % B] = c
    cats = ["human"] + ["c%d" % i for i in range(1, C)]
    glen = int(r.integers(300, 20000))
    gs = [util.random_seq(r, glen) for _ in range(B)]
    mins = []
    for g in gs:
        mins.append(np.unique(po.minimisers(g.decode(), k, w)).astype(np.uint64))
    bin_size = int(r.integers(500, 400000))
    oidx = po.Index.new(B, bin_size, b2c, cats, k=k, w=w, nhash=h)
    for b, m in enumerate(mins):
        if len(m):
            oidx.emplace_many(m, b)
    n = int(r.choice([1, 7, 64, 65, 500, 3000]))
    lmax = int(r.choice([30, 150, 1000, 5000, 30000]))
    if n * lmax > 6_000_000:
        n = max(1, 6_000_000 // lmax)
    paired = bool(r.integers(0, 2)) or C > 2
    reads = util.sample_reads(r, gs, n, (0, lmax), sub_rate=float(r.choice([0.0, 0.02, 0.1])), random_fraction=0.2)
    mates = util.sample_reads(r, gs, n, (0, min(lmax, 2000)), sub_rate=0.03) if paired else None

    def spice(s):
        a = bytearray(s)
        if len(a) > 10 and r.random() < 0.2:
            for p in r.integers(0, len(a), int(r.integers(1, 8))):
                a[int(p)] = ord("N")
        if len(a) > 200 and r.random() < 0.1:
            p = int(r.integers(0, len(a) - 150))
            a[p:p + 150] = (b"AC" * 75) if r.random() < 0.5 else b"T" * 150
        return bytes(a)
    reads = [spice(s) for s in reads]
    if mates:
        mates = [spice(s) for s in mates]
    # long reads for the SPLIT launch (single-end): stitched from random sequence, genome slices, homopolymer runs and tandem repeats, so that
    # piece seams fall into ties as well as into ordinary sequence
    split_bucket = int(r.choice([0, 64, 64, 72, 88, 255]))
    if not paired and r.random() < 0.5:
        for _ in range(int(r.integers(1, 6))):
            total, parts = int(r.integers(1000, 150000)), []
            while sum(map(len, parts)) < total:
                kind, L = int(r.integers(0, 4)), int(r.integers(50, 20000))
                if kind == 0:
                    parts.append(util.random_seq(r, L))
                elif kind == 1:
                    g = gs[int(r.integers(0, len(gs)))]
                    s0 = int(r.integers(0, max(1, len(g) - 1)))
                    parts.append(util.mutate(r, g[s0:s0 + L], 0.03))
                elif kind == 2:
                    parts.append(bytes([b"ACGTN"[int(r.integers(0, 5))]]) * min(L, 3000))
                else:
                    unit = util.random_seq(r, int(r.integers(1, 13)))
                    parts.append((unit * (min(L, 4000) // len(unit) + 1))[:min(L, 4000)])
            reads[int(r.integers(0, len(reads)))] = b"".join(parts)[:total]
        n = len(reads)
    if not any(len(s) for s in reads):
        reads[0] = b"ACGT" * 20
    if r.random() < 0.2:
        msg = sharded_trial(r, oidx, reads, mates, C, paired)
        oidx.free()
        return "k=%d w=%d B=%d C=%d h=%d S=%d n=%d lmax=%d paired=%d %s" % (k, w, B, C, h, bin_size, n, lmax, paired, msg)
    g = util.gpu_index_from_oracle(api, oidx)
    try:
        p = pack.pack_reads(reads, mates)
        tiny = bool(r.random() < 0.15)
        st = api.Stream(g, max(n, 1), p["n_bases"], tiny_log=tiny, split_bucket=split_bucket)
        st.set_model(api.default_model(C, oidx.host_index if not paired else 0, paired=paired))
        twice = bool(r.random() < 0.3)
        for _ in range(2 if twice else 1):
            st.submit_host(p, np.full(n, 40.0, np.float32), np.zeros(n, np.float32))
        outs = [st.wait_host() for _ in range(2 if twice else 1)]
        gpu = outs[0]
        st.destroy()
        seqs, offs, split = util.concat(reads, mates)
        orc = oidx.process_reads(seqs, offs, mate_split=split, mq_const=40.0, threads=8)
        for o in outs:
            util.assert_parity(o, orc)
    finally:
        g.destroy()
        oidx.free()
    return "k=%d w=%d B=%d C=%d h=%d S=%d n=%d lmax=%d longest=%d paired=%d split_bucket=%d tiny_log=%d hashes=%d" % (
        k, w, B, C, h, bin_size, n, lmax, max(map(len, reads)), paired, split_bucket, tiny, int(gpu["num_hashes"].sum()))


def sharded_trial(r, oidx, reads, mates, C, paired):
    """row-range sharded index (chn_shard_*): N shards with random cut points on one GPU, per-shard partial probe words summed on
    the host (what the RCCL sum all-reduce computes), every shard finishes with the total -> identical to the oracle"""
    S, Wd, h = oidx.bin_size, oidx.bin_words, oidx.hash_funs
    nsh = int(r.integers(2, 5))
    cuts = sorted(set([0, S] + [int(x) for x in r.integers(1, max(S, 2), nsh - 1)]))
    words = oidx.words()
    host = oidx.host_index if oidx.host_index < 255 else 255
    shards = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        d = api.make_desc(oidx.bins, S, oidx.bin_to_cat, C, host, k=oidx.k, w=oidx.w, hash_funs=h, row_begin=lo, row_end=hi)
        sh = api.Index(d)
        sh.upload(words[lo * Wd:hi * Wd], row_begin=lo)
        shards.append(sh)
    p = pack.pack_reads(reads, mates)
    n = len(reads)
    sts = [api.Stream(sh, n, p["n_bases"]) for sh in shards]
    bufs = []
    try:
        for st in sts:
            st.set_model(api.default_model(C, host if not paired else 0, paired=paired))
        Es = [st.shard_minimise_host(p) for st in sts]
        assert len(set(Es)) == 1
        nwords = max(1, Es[0] * h * Wd)
        bufs = [api.device_malloc(0, nwords * 8) for _ in shards]
        for st, sh, buf in zip(sts, shards, bufs):
            api.device_upload(0, buf, np.zeros(nwords, np.uint64))  # a batch without minimisers writes nothing into its buffer
            st.shard_probe(sh, buf, nwords)
        parts = [api.device_download(0, buf, nwords * 8, np.uint64) for buf in bufs]
        total = np.zeros(nwords, np.uint64)
        seen = np.zeros(nwords, bool)
        for q in parts:
            assert not np.any(seen & (q != 0)), "a probe word is non-zero in two shards"
            seen |= q != 0
            total += q
        seqs, offs, split = util.concat(reads, mates)
        orc = oidx.process_reads(seqs, offs, mate_split=split, mq_const=0.0, threads=8)
        for st, buf in zip(sts, bufs):
            api.device_upload(0, buf, total)
            st.shard_finish(buf)
            util.assert_parity(st.wait_host(), orc)
    finally:
        for buf in bufs:
            api.device_free(0, buf)
        for st in sts:
            st.destroy()
        for sh in shards:
            sh.destroy()
    return "row-sharded x%d E=%d" % (len(shards), Es[0])


def main():
    secs, seed = float(sys.argv[1]), int(sys.argv[2])
    po.build()
    r = np.random.default_rng(seed)
    t0, t = time.time(), 0
    while time.time() - t0 < secs:
        state = r.bit_generator.state
        try:
            msg = trial(r, t)
        except Exception as e:  # noqa: BLE001
            print("MISMATCH/ERROR in trial %d (seed %d): %r" % (t, seed, e), flush=True)
            print("rng state:", state, flush=True)
            raise
        print("trial %d ok  %s  [%.0fs]" % (t, msg, time.time() - t0), flush=True)
        t += 1
    print("fuzz_parity: %d trials, no mismatch" % t)


if __name__ == "__main__":
    main()
