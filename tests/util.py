"""Shared helpers for the parity tests: seeded synthetic genomes/reads and oracle<->GPU index mirroring."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rng(seed):
    return np.random.default_rng(seed)


def random_seq(r, n):
    return np.frombuffer(b"ACGT", np.uint8)[r.integers(0, 4, n)].tobytes()


def mutate(r, seq, rate):
    a = np.frombuffer(seq, np.uint8).copy()
    hit = r.random(a.size) < rate
    alt = np.frombuffer(b"ACGT", np.uint8)[r.integers(0, 4, a.size)]
    a[hit] = alt[hit]
    return a.tobytes()


def sample_reads(r, genomes, n, length, sub_rate=0.05, random_fraction=0.1):
    """genomes: list of bytes.  Returns list of reads (bytes)."""
    out = []
    for _ in range(n):
        L = int(length if np.isscalar(length) else r.integers(length[0], length[1] + 1))
        if r.random() < random_fraction or not genomes:
            out.append(random_seq(r, L))
        else:
            g = genomes[int(r.integers(0, len(genomes)))]
            if len(g) <= L:
                out.append(mutate(r, g, sub_rate))
            else:
                s = int(r.integers(0, len(g) - L))
                out.append(mutate(r, g[s:s + L], sub_rate))
    return out


def build_oracle_index(po, genomes_by_bin, bin_to_cat, categories, k=19, w=41, bin_size=None, fill_seed=None, fill=0.0):
    """genomes_by_bin: list (one entry per bin) of lists of sequences.  Inserts the distinct minimisers of every
    sequence into its bin (what `charon index` does, src/index_main.cpp:118-160,238-263)."""
    mins = []
    for seqs in genomes_by_bin:
        s = set()
        for g in seqs:
            s.update(int(x) for x in po.minimisers(g.decode() if isinstance(g, bytes) else g, k, w))
        mins.append(np.array(sorted(s), dtype=np.uint64))
    if bin_size is None:
        bin_size = int(po.lib().orc_bin_size_in_bits(max(1, max(len(m) for m in mins)), 3, 0.01))
    idx = po.Index.new(len(genomes_by_bin), bin_size, bin_to_cat, categories, k=k, w=w)
    if fill_seed is not None and fill > 0:
        r = rng(fill_seed)
        words = idx.words()
        B = len(genomes_by_bin)
        for wd in range(idx.bin_words):
            nb = min(64, B - 64 * wd)
            if nb <= 0:
                continue
            bits = (r.random((idx.bin_size, nb)) < fill)
            vals = np.zeros(idx.bin_size, np.uint64)
            for b in range(nb):
                vals |= bits[:, b].astype(np.uint64) << np.uint64(b)
            words[wd::idx.bin_words] |= vals
    for b, m in enumerate(mins):
        if len(m):
            idx.emplace_many(m, b)
    return idx


def gpu_index_from_oracle(api, oidx, device=0):
    host = oidx.host_index if oidx.host_index < 255 else 255
    desc = api.make_desc(oidx.bins, oidx.bin_size, oidx.bin_to_cat, oidx.ncat, host, k=oidx.k, w=oidx.w,
                         hash_funs=oidx.hash_funs, device=device)
    g = api.Index(desc)
    g.upload(oidx.words())
    return g


def concat(reads, mates=None):
    """-> (bytes, offsets uint64[n+1], mate_split or None) in the oracle's input layout"""
    parts, offs, split = [], [0], []
    for i, s in enumerate(reads):
        t = s + (mates[i] if mates is not None else b"")
        parts.append(t)
        offs.append(offs[-1] + len(t))
        split.append(len(s))
    return b"".join(parts), np.array(offs, np.uint64), (np.array(split, np.uint32) if mates is not None else None)


def assert_parity(gpu, orc, prob_tol=1e-6):
    """bit-exact integer columns; probability column within the north-star tolerance (1e-6)."""
    np.testing.assert_array_equal(gpu["num_hashes"], orc["num_hashes"])
    np.testing.assert_array_equal(gpu["counts"], orc["counts"])
    np.testing.assert_array_equal(gpu["unique"], orc["unique"])
    np.testing.assert_array_equal(gpu["conf"], orc["conf"])
    np.testing.assert_array_equal(gpu["call"], orc["call"])
    a, b = gpu["probs"], orc["probs"]
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    assert np.max(np.abs(a[ok] - b[ok]), initial=0.0) <= prob_tol


def download_results(api, res, n, ncat, device=0):
    """device-resident chn_result -> the dict layout of Stream.wait_host()"""
    return dict(num_hashes=api.device_download(device, res.num_hashes, n * 4, np.uint32),
                counts=api.device_download(device, res.counts, n * ncat * 4, np.uint32).reshape(n, ncat),
                unique=api.device_download(device, res.unique_counts, n * ncat * 4, np.uint32).reshape(n, ncat),
                probs=api.device_download(device, res.probabilities, n * ncat * 8, np.float64).reshape(n, ncat),
                call=api.device_download(device, res.call, n, np.uint8), conf=api.device_download(device, res.confidence, n, np.uint8),
                flags=api.device_download(device, res.flags, n, np.uint8))


def free_synth_reads(api, rd, device=0):
    for p in (rd.bases2, rd.seg1_offset, rd.seg1_length, rd.mean_quality, rd.compression):
        api.device_free(device, p)


def assert_same_results(a, b, keys=("num_hashes", "counts", "unique", "call", "conf", "probs")):
    for key in keys:
        assert np.array_equal(a[key], b[key], equal_nan=True) if key == "probs" else np.array_equal(a[key], b[key]), key
