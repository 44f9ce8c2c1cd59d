"""CPU test of the N > 1 path: two gloo ranks each classify their shard of one read set (with the CPU oracle standing in
for the device, since there is no GPU here), merge the summary counters with the same all-reduce bench.py uses, and
must reproduce the single-process result.  The data path itself has no collective (reads are independent units)."""
import os
import tempfile

import numpy as np
import pytest

from tests import util


def _worker(rank, world, initfile, outdir):
    import torch.distributed as dist
    from charon_amd import shard
    from oracle import pyoracle as po
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    r = util.rng(77)
    gs = [util.random_seq(r, 5000), util.random_seq(r, 5000)]
    oidx = util.build_oracle_index(po, [[g] for g in gs], [0, 1], ["host", "microbial"])
    reads = util.sample_reads(r, gs, 101, (200, 600))  # odd count: shards differ in size
    lo, hi = shard.shard_range(len(reads), rank, world)
    seqs, offs, _ = util.concat(reads[lo:hi])
    o = oidx.process_reads(seqs, offs)
    merged = shard.merge_summary(shard.summary_counts(o["call"], 2), dist)
    np.save(os.path.join(outdir, "r%d.npy" % rank), np.concatenate([[lo, hi], merged]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_summary_merge(oracle_lib):
    import torch.multiprocessing as mp
    from charon_amd import shard
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_worker, args=(2, initfile, d), nprocs=2, join=True)
        a, b = np.load(os.path.join(d, "r0.npy")), np.load(os.path.join(d, "r1.npy"))
    assert (a[0], a[1], b[0], b[1]) == (0, 51, 51, 101)
    assert np.array_equal(a[2:], b[2:])
    # single-process reference
    r = util.rng(77)
    gs = [util.random_seq(r, 5000), util.random_seq(r, 5000)]
    oidx = util.build_oracle_index(oracle_lib, [[g] for g in gs], [0, 1], ["host", "microbial"])
    reads = util.sample_reads(r, gs, 101, (200, 600))
    seqs, offs, _ = util.concat(reads)
    want = shard.summary_counts(oidx.process_reads(seqs, offs)["call"], 2)
    assert np.array_equal(a[2:], want) and want.sum() == 101
    oidx.free()


def test_shard_range_properties():
    from charon_amd import shard
    for total in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


def _shard_worker(rank, world, initfile, outdir):
    """row-sharded mode on CPU: each rank fills partial[e][i][w] only for rows it owns; the gloo SUM all-reduce must
    reconstruct every probe word exactly (one owner per row), and the AND over hash functions must equal bulk_contains"""
    import torch
    import torch.distributed as dist
    from charon_amd import shard
    from oracle import pyoracle as po
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    r = util.rng(5)
    gs = [util.random_seq(r, 3000) for _ in range(70)]
    oidx = util.build_oracle_index(po, [[g] for g in gs], [i % 2 for i in range(70)], ["host", "microbial"], bin_size=5003)
    words, W, S = oidx.words(), oidx.bin_words, oidx.bin_size
    mins = np.concatenate([po.minimisers(g[:800].decode()) for g in gs[:6]])
    lo, hi = shard.shard_range(S, rank, world)
    partial = np.zeros((len(mins), 3, W), dtype=np.int64)
    for e, v in enumerate(mins):
        for i in range(3):
            row = po.lib().orc_hash_and_fit(int(v), i, S)
            if lo <= row < hi:
                partial[e, i] = words[row * W:(row + 1) * W].view(np.int64)
    t = torch.from_numpy(partial)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    full = t.numpy().view(np.uint64)
    rows = full[:, 0] & full[:, 1] & full[:, 2]
    want = np.stack([oidx.bulk_contains(int(v)) for v in mins])
    ok = bool(np.array_equal(rows, want))
    np.save(os.path.join(outdir, "s%d.npy" % rank), np.array([int(ok), lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_probe_words_sum_reconstructs(oracle_lib):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_shard_worker, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        a, b = np.load(os.path.join(d, "s0.npy")), np.load(os.path.join(d, "s1.npy"))
    assert a[0] == 1 and b[0] == 1
    assert (a[1], a[2], b[1], b[2]) == (0, 2502, 2502, 5003)


def test_bench_launches_its_own_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` without torch.distributed.run around it must start the two ranks itself (before touching the
    GPU), rendezvous on 127.0.0.1, reduce the timing with MAX over ranks and relay exactly one JSON line from rank 0.  There is
    no GPU here, so the steps are rehearsed (--launch-check); the launch, rendezvous and reduction code is the real one."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check", "--steps", "2"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["rehearsal"] is True and out["launch_check"] is True
    # rank 1 sleeps twice as long as rank 0: the reported time is the MAX over ranks
    assert out["ms_per_step"] >= 19.0
    # more ranks than GPUs on the RCCL backend is refused outright (no GPU here -> any N > 0 is too many)
    p = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode != 0 and "RCCL needs one GPU per rank" in (p.stderr + p.stdout)


def _sparse_worker(rank, world, initfile, outdir):
    """sparse row-sharded exchange on CPU: each rank owns a row range AND its own minimisers; queries grouped by owner go out
    through shard.all_to_all_v, the owner gathers its rows, the rows come back through the reverse exchange in query order, and
    the AND over the hash functions must equal bulk_contains on the whole index"""
    import torch
    import torch.distributed as dist
    from charon_amd import shard
    from oracle import pyoracle as po
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    r = util.rng(5)
    gs = [util.random_seq(r, 3000) for _ in range(70)]
    oidx = util.build_oracle_index(po, [[g] for g in gs], [i % 2 for i in range(70)], ["host", "microbial"], bin_size=5003)
    words, W, S = oidx.words().reshape(-1, oidx.bin_words), oidx.bin_words, oidx.bin_size
    splits = shard.row_splits(S, world)
    mins = np.concatenate([po.minimisers(g[:700].decode()) for g in gs[rank * 5:rank * 5 + 5 + rank]])  # ranks hold different amounts
    rows = np.array([[po.lib().orc_hash_and_fit(int(v), i, S) for i in range(3)] for v in mins], dtype=np.int64)
    owner = np.searchsorted(np.array(splits[1:]), rows, side="right")
    order = np.lexsort((np.arange(rows.size), owner.ravel()))          # grouped by owner, original probe order inside a group
    send = torch.from_numpy((rows.ravel()[order] - np.array(splits)[owner.ravel()[order]]).astype(np.int64))
    send_counts = [int((owner == o).sum()) for o in range(world)]
    qin, recv_counts = shard.all_to_all_v(dist, send, send_counts)
    lo = splits[rank]
    served = torch.from_numpy(words[lo + qin.numpy()].astype(np.int64).ravel())   # the owner's gather
    back, back_counts = shard.all_to_all_v(dist, served, recv_counts, width=W)
    ok = back_counts == send_counts
    got = np.zeros((rows.size, W), np.uint64)
    got[order] = back.numpy().view(np.uint64).reshape(-1, W)
    got = got.reshape(len(mins), 3, W)
    anded = got[:, 0] & got[:, 1] & got[:, 2]
    want = np.stack([oidx.bulk_contains(int(v)) for v in mins])
    ok = ok and bool(np.array_equal(anded, want))
    np.save(os.path.join(outdir, "x%d.npy" % rank), np.array([int(ok), len(mins)]))
    dist.barrier()
    dist.destroy_process_group()


def test_sparse_row_sharded_exchange_two_ranks(oracle_lib):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_sparse_worker, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        a, b = np.load(os.path.join(d, "x0.npy")), np.load(os.path.join(d, "x1.npy"))
    assert a[0] == 1 and b[0] == 1 and a[1] != b[1]
