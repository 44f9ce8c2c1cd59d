"""GPU parity tests (-m gpu) of long reads cut over the 64 lanes of a wavefront (the reference caps a read's length
only at uint32, src/dehost_main.cpp:346-350, and minimises it sequentially, :367-370).

A single-end read of 32 768 bases or more is rolled by one wavefront, lane j taking piece j; pieces overlap by w - 1 bases and a piece
other than the first starts "cold" -- exact only where the window minimum is unique, so a piece whose first window has a tie is not
started and its predecessor runs on through it.  Everything must stay bit-exact against the oracle's sequential rule: num_hashes,
counts, unique counts, call, confidence (probabilities within 1e-6)."""
import time

import numpy as np
import pytest

from tests import util
from tests.test_gpu_parity import run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import charon_amd.api as api
    return api


def _to_device(api, arr):
    arr = np.ascontiguousarray(arr)
    p = api.device_malloc(0, max(arr.nbytes, 16))
    api.device_upload(0, p, arr)
    return p


def _run(api, gidx, reads, split_bucket=0, tiny_log=False, twice=False):
    from charon_amd import pack
    p = pack.pack_reads(reads)
    n = len(reads)
    st = api.Stream(gidx, n, p["n_bases"], tiny_log=tiny_log, split_bucket=split_bucket)
    st.set_model(api.default_model(gidx.desc.num_categories, gidx.desc.host_index))
    mq, cp = np.full(n, 40.0, np.float32), np.zeros(n, np.float32)
    st.submit_host(p, mq, cp)
    if twice:
        st.submit_host(p, mq, cp)
    outs = [st.wait_host() for _ in range(2 if twice else 1)]
    reruns = st.profile(4)[1]
    st.destroy()
    return outs, reruns


def _awkward_reads(r, gs, scale):
    """reads that stress the seams: `scale` = 1 for the default limit (32 768 bases), smaller for a lowered one"""
    L = lambda x: max(200, int(x * scale))
    reads = [util.mutate(r, gs[0][1000:1000 + L(200000)], 0.05),                      # long, mostly host
             util.mutate(r, gs[1][5:5 + L(120005)], 0.1),
             b"A" * L(100000),                                                         # homopolymer: every seam is a tie -> one lane runs on through all pieces
             b"ACGT" * (L(100000) // 4),                                               # tandem repeats: ties at every seam
             b"ACGGTCA" * (L(70000) // 7),
             util.random_seq(r, L(40000)) + b"T" * L(30000) + util.random_seq(r, L(50000)),   # ties only at some seams: pieces merge locally
             (b"ACGTTGCA" * 40 + util.random_seq(r, 700)) * (L(90000) // 1020),       # repeats that come and go
             util.random_seq(r, L(32768)), util.random_seq(r, L(32768) - 1), util.random_seq(r, L(32768) + 1),   # around the limit
             util.random_seq(r, L(36863)), util.random_seq(r, L(36864)), util.random_seq(r, L(65535)), util.random_seq(r, L(65537))]
    n_run = util.random_seq(r, L(50000))
    n_run = n_run[:L(20000)] + b"N" * 100 + n_run[L(20000) + 100:L(31000)] + b"NNNNNNN" + n_run[L(31000) + 7:]
    reads.append(n_run)                                                                # N runs (an N mask in the batch)
    return reads


def _index_pair(oracle_lib, r, glen):
    gs = [util.random_seq(r, glen), util.random_seq(r, glen)]
    fused = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "microbial"])            # 2 bins, W = 1: LDS counters
    many = [gs[i % 2][(i // 2) * (glen // 35):(i // 2 + 1) * (glen // 35) + 40] for i in range(70)]
    rows = util.build_oracle_index(oracle_lib, [[m] for m in many], [i % 2 for i in range(70)], ["human", "microbial"], fill_seed=8, fill=0.08)  # 70 bins, W = 2: row log
    return gs, fused, rows


def test_long_reads_over_the_lanes_equal_the_sequential_rule(api, oracle_lib):
    r = util.rng(2024)
    gs, fused, rows = _index_pair(oracle_lib, r, 1200000)
    reads = _awkward_reads(r, gs, 1.0) + [util.mutate(r, gs[0][:1000000], 0.03), util.mutate(r, gs[1][100000:1148576], 0.08)] + \
        util.sample_reads(r, gs, 150, (100, 31000))
    reads[-3] = b""
    for oidx in (fused, rows):
        orc = run_oracle(oidx, reads)
        g = util.gpu_index_from_oracle(api, oidx)
        (out,), _ = _run(api, g, reads)
        util.assert_parity(out, orc)
        (ref,), _ = _run(api, g, reads, split_bucket=255)      # one lane per read, as before
        util.assert_same_results(out, ref)
        g.destroy()
    assert orc["num_hashes"][0] > 15000 and orc["num_hashes"][2] == 4347   # homopolymer (tie path): the first window, then one emission per 23 values
    fused.free(); rows.free()


@pytest.mark.parametrize("kw", [(19, 41), (15, 15), (11, 60), (27, 27), (5, 200)])
def test_lowered_limit_splits_ordinary_reads(api, oracle_lib, kw):
    """CHN_STREAM_SPLIT_BUCKET(64): every read of 1 024 bases or more is cut over a wavefront -- many reads, few pieces each, other
    (k, w), two batches in flight, the overflow re-run"""
    k, w = kw
    r = util.rng(7 + k)
    gs = [util.random_seq(r, 40000), util.random_seq(r, 40000)]
    fused = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "microbial"], k=k, w=w)
    # (slices that do not overlap and a light random fill: few rows with more than three set bins, so that the density-sized log of the
    #  first run holds them even at one minimiser per base -- the re-run path is what `tiny` is for)
    many = [gs[i % 2][(i // 2) * 1100:(i // 2 + 1) * 1100] for i in range(70)]
    rows = util.build_oracle_index(oracle_lib, [[m] for m in many], [i % 2 for i in range(70)], ["human", "microbial"], k=k, w=w, fill_seed=3, fill=0.02)
    reads = _awkward_reads(r, gs, 1 / 16.0) + util.sample_reads(r, gs, 400, (50, 6000), sub_rate=0.04) + [b"", b"ACG", b"T" * 1024, b"GA" * 3000]
    for oidx, tiny in ((fused, False), (rows, False), (rows, True)):
        orc = run_oracle(oidx, reads)
        g = util.gpu_index_from_oracle(api, oidx)
        outs, reruns = _run(api, g, reads, split_bucket=64, tiny_log=tiny, twice=True)
        for o in outs:
            util.assert_parity(o, orc)
        # (k = 5: a few hundred distinct canonical k-mers and every bin holds them all -- rows with many set bins overrun the density-sized
        #  log legitimately and the batch is re-run)
        assert reruns == 2 if tiny else (reruns == 0 or k == 5)
        g.destroy()
    fused.free(); rows.free()


def test_device_batch_with_long_reads(api, oracle_lib):
    """a device-resident batch: the host cannot count the long reads, the launch is sized by n_bases / limit"""
    from charon_amd import pack
    r = util.rng(31)
    gs, fused, rows = _index_pair(oracle_lib, r, 300000)
    reads = util.sample_reads(r, gs, 300, (500, 9000)) + [util.mutate(r, gs[0][:250000], 0.05), b"C" * 40000, util.random_seq(r, 33000)]
    p = pack.pack_reads(reads)
    n = len(reads)
    dev = [_to_device(api, np.ascontiguousarray(p[key], dt)) for key, dt in (("bases2", np.uint32), ("seg1_offset", np.uint64), ("seg1_length", np.uint32))]
    mq = _to_device(api, np.full(n, 40.0, np.float32))
    for oidx in (fused, rows):
        orc = run_oracle(oidx, reads)
        g = util.gpu_index_from_oracle(api, oidx)
        for bucket in (0, 64):
            st = api.Stream(g, n, p["n_bases"], split_bucket=bucket)
            st.set_model(api.default_model(2, g.desc.host_index))
            st.submit_device(n, p["n_bases"], dev[0], dev[1], dev[2], mq, None)
            out = util.download_results(api, st.wait_device(), n, 2)
            util.assert_parity(out, orc)
            st.destroy()
        g.destroy()
    for d in dev + [mq]:
        api.device_free(0, d)
    fused.free(); rows.free()


def test_one_megabase_read_does_not_hold_up_a_batch(api, oracle_lib):
    """a batch of 65 536 reads of 10 kb plus ONE read of 1 Mb must cost at most 1.5 x the batch without it
    (one lane per read: the launch lasts as long as that lane, some 0.2 s against 4 ms)."""
    from charon_amd import pack
    B, S = 2, 1 << 24
    g = api.Index(api.make_desc(B, S, [0, 1], 2, 0))
    n_gen, glen = 2, 1 << 20
    gen = api.synth_genomes(0, 43, n_gen, glen)
    g.synth_fill(43, 0.05)
    g.synth_plant(gen, n_gen, glen, [0, 1])
    n = 65536
    rd = api.synth_reads(0, 42, gen, n_gen, glen, n, 10000, 10000, 0.05, 0.1, 40.0)
    bases = api.device_download(0, rd.bases2, rd.n_bases // 4, np.uint32)
    offs = api.device_download(0, rd.seg1_offset, n * 8, np.uint64)
    lens = api.device_download(0, rd.seg1_length, n * 4, np.uint32)
    long_read = util.random_seq(util.rng(5), 1000000)
    pl = pack.pack_reads([long_read])
    bases2 = np.concatenate([bases, np.ascontiguousarray(pl["bases2"], np.uint32)])
    offs2 = np.concatenate([offs, np.array([rd.n_bases], np.uint64)])
    lens2 = np.concatenate([lens, np.array([1000000], np.uint32)])
    nb2 = rd.n_bases + pl["n_bases"]
    d_b, d_o, d_l = _to_device(api, bases2), _to_device(api, offs2), _to_device(api, lens2)
    mq = _to_device(api, np.full(n + 1, 40.0, np.float32))

    def timed(nr, nb, split_bucket):
        st = api.Stream(g, n + 1, nb2, split_bucket=split_bucket)
        st.set_model(api.default_model(2, 0))
        best, res = 1e9, None
        for _ in range(6):
            t0 = time.perf_counter()
            st.submit_device(nr, nb, d_b, d_o, d_l, mq, None)
            res = st.wait_device()
            best = min(best, time.perf_counter() - t0)
        out = util.download_results(api, res, nr, 2)
        st.destroy()
        return best, out

    t_without, out_a = timed(n, rd.n_bases, 0)
    t_with, out_b = timed(n + 1, nb2, 0)
    if t_with > 1.4 * t_without:  # (a noisy neighbour on the box: time both once more and keep the better figures)
        t_without = min(t_without, timed(n, rd.n_bases, 0)[0])
        t_with = min(t_with, timed(n + 1, nb2, 0)[0])
    t_one_lane, out_c = timed(n + 1, nb2, 255)
    print("65 536 x 10 kb: %.2f ms; + one 1 Mb read: %.2f ms (x %.2f); the same with one lane per read: %.2f ms" %
          (t_without * 1e3, t_with * 1e3, t_with / t_without, t_one_lane * 1e3))
    util.assert_same_results(out_b, out_c)
    for key in ("num_hashes", "counts", "unique", "call"):
        assert np.array_equal(out_a[key], out_b[key][:n]), key
    assert out_b["num_hashes"][n] > 80000
    assert t_with <= 1.5 * t_without, (t_with, t_without)
    for d in (d_b, d_o, d_l, mq, gen):
        api.device_free(0, d)
    util.free_synth_reads(api, rd)
    g.destroy()
