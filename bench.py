#!/usr/bin/env python3
"""bench.py -- throughput of the `charon dehost` per-read classification hot path on MI355X.

One "step" = one pass of the whole device chain (length ordering -> minimise+probe -> count -> KDE model+call)
over one batch of synthetic reads that is already resident in HBM.  Workloads:

  39g   (default) 5 kb reads vs the 39 GB stand-in index: B=100 bins (50 human / 50 microbial), TB=128, W=2,
        S=2 437 500 000 rows, background fill 21.5 %, one planted genome per bin; reads 45/45/10 % host / microbial /
        random with 5 % substitutions.  This is the configuration BASELINE.json's metric is quoted on.
  cfg2  5 kb reads vs the 2-category 1 GiB index (B=2, W=1, S=2^27)  -- BASELINE configs[1]
  small tiny smoke-sized variant of cfg2 (CPU-container rehearsal of the harness is impossible: needs a GPU)

Multi-GPU (`--gpus N`): the 39 GB index fits every GPU, so the path shards by READ with a full index replica per rank
and no data-path collective ("weak" scaling: per-GPU batch fixed).  The only collective is the barrier / max-reduce of
the timing contract and one all-reduce of the summary counters.  Started under torch.distributed.run (WORLD_SIZE set)
this process is one rank; started plainly with --gpus N > 1 it launches the N ranks itself (before touching the GPU)
and relays rank 0's JSON line.

What is timed: the device chain on packed reads that are ALREADY RESIDENT IN HBM (`config.input`); FASTQ parsing, the
mean-quality and gzip-ratio columns are host work outside the timed region.  The host-buffer entry (H2D of the packed
batch + D2H of the results, pinned memory, three batches in flight) is timed separately and reported as
`config.pcie_inclusive_reads_per_s` -- it is never `value`.  Steps cycle through `--read-sets` distinct read sets so that no
step replays the previous step's batch (the model kernel's memo table sees new keys, as in a real run).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)

WORKLOADS = {
    # name: bins, rows, genome_len per bin, description
    "39g": dict(bins=100, rows=2437500000, genome_len=1 << 22, reads=1 << 20, read_len=5000, fill=0.215,
                desc="5 kb synthetic reads vs 39 GB stand-in index (B=100, TB=128, W=2, S=2437500000, h=3, k=19, w=41)"),
    "cfg2": dict(bins=2, rows=1 << 27, genome_len=1 << 24, reads=1 << 20, read_len=5000, fill=0.215,
                 desc="5 kb synthetic reads vs 2-category 1 GiB index (B=2, TB=64, W=1, S=2^27, h=3, k=19, w=41)"),
    # BASELINE configs[4]: paired short reads (2 x 150 b) vs an 8-category index (one bin per category, one named "host")
    "cfg5": dict(bins=8, rows=1 << 27, genome_len=1 << 23, reads=1 << 23, read_len=150, fill=0.215, paired=True,
                 desc="paired 2 x 150 b synthetic reads vs 8-category 1 GiB index (B=8, TB=64, W=1, S=2^27, h=3, k=19, w=41)"),
    "small": dict(bins=2, rows=1 << 20, genome_len=1 << 16, reads=1 << 14, read_len=1000, fill=0.1,
                  desc="1 kb synthetic reads vs toy 2-category index (harness check)"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="39g", choices=sorted(WORKLOADS))
    ap.add_argument("--reads-per-step", type=int, default=0, help="per GPU; default from the workload")
    ap.add_argument("--read-len", type=int, default=0)
    ap.add_argument("--read-len-max", type=int, default=0, help="log-uniform lengths in [read-len, read-len-max] (BASELINE configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; nccl = RCCL (default). gloo only to rehearse N ranks on fewer GPUs")
    ap.add_argument("--shard", default="reads", choices=["reads", "rows", "rows-dense"],
                    help="reads (default): full index replica per rank, reads sharded, no data-path collective; "
                         "rows: index row-range sharded over ranks (for indexes that do not fit one GPU), every rank classifies its own "
                         "reads and exchanges row queries / row words with the owners (two all-to-alls per batch); "
                         "rows-dense: the older scheme, every rank minimises the same batch, one RCCL sum all-reduce of all probe words")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer leg (H2D of the packed batch + D2H of the results)")
    ap.add_argument("--inflight", type=int, default=3, help="device batches in flight in the timed loop (1..3)")
    ap.add_argument("--read-sets", type=int, default=3, help="distinct synthetic read sets the steps cycle through")
    ap.add_argument("--launch-check", action="store_true",
                    help="rehearse the launch / rendezvous / timing-reduction plumbing only (no GPU work; CPU test of --gpus N)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "WORLD_SIZE" in os.environ  # launched by torch.distributed.run (also with one rank: same RCCL code path)
    ndev = torch.cuda.device_count()       # does not initialise the GPU
    if args.launch_check:
        return launch_check(args, rank, world, use_dist)
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the hot path)")
    if args.backend == "nccl" and world > ndev:
        raise SystemExit("bench.py: %d ranks on %d GPU(s) -- RCCL needs one GPU per rank (use --backend gloo only to rehearse the "
                         "launch on fewer GPUs; such a run is labelled \"rehearsal\" and is not a scaling figure)" % (world, ndev))
    rehearsal = world > ndev or (args.backend == "gloo" and world > 1)
    dev = local % ndev  # == local on a real N-GPU launch; wraps only in a gloo rehearsal on fewer GPUs
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
    coll_dev = torch.device("cuda", dev) if args.backend == "nccl" else torch.device("cpu")
    local = dev
    torch.cuda.set_device(local)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    import charon_amd.api as api
    from charon_amd import shard
    wl = dict(WORKLOADS[args.workload])
    if args.reads_per_step:
        wl["reads"] = args.reads_per_step
    if args.read_len:
        wl["read_len"] = args.read_len
    B, S, n_reads, L = wl["bins"], wl["rows"], wl["reads"], wl["read_len"]
    paired = bool(wl.get("paired"))
    if paired:
        b2c = list(range(B))          # one category per bin, category 0 = host
        categories = ["host"] + ["c%d" % i for i in range(1, B)]
    else:
        b2c = [b % 2 for b in range(B)]  # even bins: category 0 (human), odd bins: category 1 (microbial)
        categories = ["human", "microbial"]
    ncat = len(categories)

    t_setup = time.time()
    rows_mode = args.shard == "rows-dense"   # every rank works on the SAME batch
    sparse_mode = args.shard == "rows"      # every rank works on its own reads, rows are fetched from their owners
    if rows_mode or sparse_mode:
        rlo, rhi = shard.shard_range(S, rank, world)
        desc = api.make_desc(B, S, b2c, ncat, 0, device=local, row_begin=rlo, row_end=rhi)
    else:
        desc = api.make_desc(B, S, b2c, ncat, 0, device=local)
    index = api.Index(desc)
    index.synth_fill(43, wl["fill"])
    genomes = api.synth_genomes(local, 43, B, wl["genome_len"])
    index.synth_plant(genomes, B, wl["genome_len"], list(range(B)))
    # Weak scaling: rank r classifies global reads [r*n, (r+1)*n) of read set j, and set j starts at global read j*world*n of one
    # seeded stream (a read's content depends only on (seed, global index): charon_amd/shard.py), so every step of every rank
    # works on reads no other step or rank has seen.  Row-sharded mode: every rank gets the SAME batch (lo = 0).
    Lmax = max(L, args.read_len_max)
    nseg = 2 if paired else 1   # mates are consecutive synthetic reads 2i, 2i+1 of one read set
    n_sets = max(1, min(args.read_sets, args.steps + args.warmup))
    sets = []
    for j in range(n_sets):
        lo = j * world * n_reads + (0 if rows_mode else rank * n_reads)
        rd = api.synth_reads(local, 42, genomes, B, wl["genome_len"], n_reads * nseg, L, Lmax, 0.05, 0.10, 40.0, first_read_id=lo * nseg)
        seg = {}
        if paired:
            off = api.device_download(local, rd.seg1_offset, n_reads * 2 * 8, np.uint64)
            ln = api.device_download(local, rd.seg1_length, n_reads * 2 * 4, np.uint32)
            for name, arr in (("o1", off[0::2]), ("o2", off[1::2]), ("l1", ln[0::2]), ("l2", ln[1::2])):
                seg[name] = api.device_malloc(local, arr.nbytes)
                api.device_upload(local, seg[name], np.ascontiguousarray(arr))
        sets.append((rd, seg))
    max_bases = max(int(rd.n_bases) for rd, _ in sets)
    stream = api.Stream(index, n_reads, max_bases, profile=True)
    stream.set_model(api.default_model(ncat, 0, paired=paired))
    stream_b = None
    if sparse_mode:  # a second stream: batch i+1 is minimised while batch i's rows are fetched
        stream_b = api.Stream(index, n_reads, max_bases, profile=True)
        stream_b.set_model(api.default_model(ncat, 0, paired=paired))
    setup_s = time.time() - t_setup
    step_no = [0]  # steps submitted so far: step i uses read set i mod n_sets

    def submit(stream=stream):
        reads, seg = sets[step_no[0] % n_sets]
        step_no[0] += 1
        if paired:
            stream.submit_device(n_reads, reads.n_bases, reads.bases2, seg["o1"], seg["l1"], reads.mean_quality, reads.compression,
                                 seg2_offset=seg["o2"], seg2_length=seg["l2"])
        else:
            stream.submit_device(n_reads, reads.n_bases, reads.bases2, reads.seg1_offset, reads.seg1_length, reads.mean_quality,
                                 reads.compression)

    def run_steps_rows(k):
        """dense row-sharded chain: minimise (all ranks, same batch) -> probe own rows -> ONE sum all-reduce -> AND/count/call"""
        res = None
        for _ in range(k):
            reads, _ = sets[step_no[0] % n_sets]
            step_no[0] += 1
            e = stream.shard_minimise_device(n_reads, reads.n_bases, reads.bases2, reads.seg1_offset, reads.seg1_length,
                                             reads.mean_quality, reads.compression)
            nwords = e * 3 * ((B + 63) // 64)
            partial = torch.empty(max(nwords, 1), dtype=torch.int64, device=torch.device("cuda", local))
            stream.shard_probe(index, partial.data_ptr(), nwords)
            if use_dist:
                if args.backend == "nccl":
                    dist.all_reduce(partial, op=dist.ReduceOp.SUM)
                else:  # gloo rehearsal: through host memory
                    hp = partial.cpu()
                    dist.all_reduce(hp, op=dist.ReduceOp.SUM)
                    partial.copy_(hp)
                torch.cuda.synchronize()
            stream.shard_finish(partial.data_ptr())
            res = stream.wait_device()
            del partial
        return res

    splits = shard.row_splits(S, world)
    Wd = (B + 63) // 64
    xbuf = {}  # per stream: torch buffers of the exchange, grown on demand

    def xtensor(key, count, dtype):
        t = xbuf.get(key)
        if t is None or t.numel() < count:
            t = torch.empty(int(count * 1.1) + 1024, dtype=dtype, device=torch.device("cuda", local))
            xbuf[key] = t
        return t

    def sparse_minimise(st):
        reads, _ = sets[step_no[0] % n_sets]
        step_no[0] += 1
        st.shardx_minimise_device(n_reads, reads.n_bases, reads.bases2, reads.seg1_offset, reads.seg1_length, reads.mean_quality, reads.compression)

    def sparse_rest(st, tag, before_serve=None):
        """queries -> (all-to-all) -> serve -> (all-to-all back) -> AND / count / call.  before_serve: called once this batch's query
        kernels are queued and before its row fetches are -- the caller queues the NEXT batch's minimise kernel there, so that this
        VALU-bound kernel runs beside the HBM-bound row fetches instead of beside the short count / scatter kernels (which it slowed
        3.5 x while the row fetches then ran alone: profiles/r02/timeline_rows_sparse.txt)"""
        n_probes, counts = st.shardx_counts(splits)
        q = xtensor((tag, "q"), n_probes, torch.int32)
        st.shardx_queries(q.data_ptr(), q.numel())
        if before_serve is not None:
            if not os.environ.get("CHARON_BENCH_SPARSE_NOSYNC"):
                st.sync()  # the query kernels have run: the minimise kernel queued next starts beside the row fetches, not beside them
            before_serve()
        if use_dist and world > 1:
            st.sync()
            if args.backend == "nccl":
                qin, rc = shard.all_to_all_v(dist, q[:n_probes], counts)
            else:  # gloo rehearsal: through host memory
                qh, rc = shard.all_to_all_v(dist, q[:n_probes].cpu(), counts)
                qin = qh.to(q.device)
            n_in = sum(rc)
            torch.cuda.synchronize()
        else:
            qin, n_in, rc = q, n_probes, counts
        rows = xtensor((tag, "rows"), n_in * Wd, torch.int64)
        st.shardx_serve(index, qin.data_ptr(), n_in, rows.data_ptr())
        if use_dist and world > 1:
            st.sync()
            if args.backend == "nccl":
                back, _ = shard.all_to_all_v(dist, rows[:n_in * Wd], rc, width=Wd)
            else:
                bh, _ = shard.all_to_all_v(dist, rows[:n_in * Wd].cpu(), rc, width=Wd)
                back = bh.to(q.device)
            torch.cuda.synchronize()
            xbuf[(tag, "back")] = back  # keep alive until the batch is waited for
        else:
            back = rows
        st.shardx_finish(back.data_ptr())

    def run_steps_sparse(k):
        sts = [stream, stream_b]
        res = None
        sparse_minimise(sts[0])
        for i in range(k):
            nxt = (lambda j=i: sparse_minimise(sts[(j + 1) % 2])) if i + 1 < k else None
            if os.environ.get("CHARON_BENCH_SPARSE_EARLY") and nxt is not None:  # diagnostic: the former order
                nxt(); nxt = None
            sparse_rest(sts[i % 2], i % 2, nxt)
            res = sts[i % 2].wait_device()
        return res

    def run_steps(k):
        if rows_mode:
            return run_steps_rows(k)
        if sparse_mode:
            return run_steps_sparse(k)
        """k whole passes of the chain; --inflight batches in flight (default 3, what the library allows): step i's count and
        model+call kernels (side stream) overlap step i+1's minimise+probe kernel, and step i+2 is already queued behind that kernel
        when the host wakes up from its wait for step i (with two in flight the device idles from the end of step i's side-stream
        kernels until the host has submitted the next batch: 0.6 ms per step on 5 kb reads, 4.6 ms on 500 b - 50 kb reads).  Every
        step's work starts and ends inside the caller's timed region."""
        res = None
        depth = max(1, min(args.inflight, 3, k))
        for _ in range(depth - 1):
            submit()
        for i in range(k):
            if i + depth - 1 < k:
                submit()
            res = stream.wait_device()
        return res

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run_steps(args.warmup)
    for w in range(4):
        stream.profile(w, reset=True)
    barrier()
    t0 = time.perf_counter()
    res = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    last_reads = sets[(step_no[0] - 1) % n_sets][0]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Sub-millisecond steps (BASELINE configs[4]'s small batches): the HIP events that bracket every kernel for the roofline figures cost a
    # measurable part of such a step, so the same steps are run once more on a stream WITHOUT those events and that time is reported
    # beside ms_per_step (never instead of it).
    ms_no_events = None
    if elapsed / max(1, args.steps) < 2e-3 and not rows_mode and not sparse_mode:
        plain = api.Stream(index, n_reads, max_bases, profile=False)
        plain.set_model(api.default_model(ncat, 0, paired=paired))
        def plain_steps(k):
            depth = max(1, min(args.inflight, 3, k))  # (per call: every batch submitted here is waited for here)
            for _ in range(depth - 1):
                submit(plain)
            for i in range(k):
                if i + depth - 1 < k:
                    submit(plain)
                plain.wait_device()
        plain_steps(max(1, args.warmup))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        plain_steps(args.steps)
        torch.cuda.synchronize()
        ms_no_events = (time.perf_counter() - t1) * 1e3 / args.steps
        plain.destroy()

    # summary counters (ResultSummary, include/result.hpp:18-25): the one collective of the read-sharded mode
    call = api.device_download(local, res.call, n_reads, np.uint8)
    summary = shard.summary_counts(call, ncat)
    if use_dist and not rows_mode:  # in rows mode every rank already holds the calls of the whole batch
        summary = shard.merge_summary(summary, dist, coll_dev)

    ranks_seen = 1
    if use_dist:  # every rank that reached this point counts itself
        t = torch.ones(1, dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ranks_seen = int(t.item())

    k1_ms, k1_n = stream.profile(0)
    k2_ms, k2_n = stream.profile(1)
    k3_ms, k3_n = stream.profile(2)
    chain_ms, chain_n = stream.profile(3)
    _, reruns = stream.profile(4)
    _, fetches = stream.profile(5)   # row fetches the last batch's probe kernel issued
    alg_bytes, total_min = stream.last_batch_bytes()
    flags = api.device_download(local, res.flags, n_reads, np.uint8)

    pcie_rate = None
    if not args.no_pcie and world == 1 and not paired and not rows_mode and not sparse_mode:
        # host-buffer entry: packed batch in PINNED host memory (chn_host_alloc), three batches in flight, so the upload of batch
        # i+2 (copy stream) runs under the probe kernel of batch i+1 while batch i's count / model+call kernels finish beside it;
        # results are downloaded in stream and come back into ordinary numpy arrays
        reads = last_reads

        def pinned_copy(dev_ptr, count, dtype):
            a = api.pinned_array(count, dtype)
            a[:] = api.device_download(local, dev_ptr, count * np.dtype(dtype).itemsize, dtype)
            return a
        packed = dict(bases2=pinned_copy(reads.bases2, reads.n_bases // 16, np.uint32), nmask=None,
                      seg1_offset=pinned_copy(reads.seg1_offset, n_reads, np.uint64),
                      seg1_length=pinned_copy(reads.seg1_length, n_reads, np.uint32), n_bases=reads.n_bases)
        mqh = api.pinned_array(n_reads, np.float32)
        cph = api.pinned_array(n_reads, np.float32)
        mqh[:] = 40.0
        cph[:] = 0.3
        stream.submit_host(packed, mqh, cph)
        stream.wait_host()
        k_pcie = 20  # whole leg timed, pipeline fill (one upload before the first kernel) and drain included
        t1 = time.perf_counter()
        stream.submit_host(packed, mqh, cph)
        stream.submit_host(packed, mqh, cph)
        for i in range(k_pcie):
            if i + 2 < k_pcie:
                stream.submit_host(packed, mqh, cph)
            stream.wait_host()
        pcie_rate = k_pcie * n_reads / (time.perf_counter() - t1)
        for a in (packed["bases2"], packed["seg1_offset"], packed["seg1_length"], mqh, cph):
            api.host_free(a)

    # informational (never `value`): the same device-resident batches with the gzip `compression` column ALSO computed on the device (deflate
    # pass + tree arithmetic per read, src/utils.cpp:114-124) -- what the drop-in binary asks of the GPU per batch
    gzip_rate = None
    if not args.no_pcie and world == 1 and not paired and not rows_mode and not sparse_mode and Lmax <= 61440:
        reads = last_reads
        k_gz = 3

        def gz_batch():
            stream.submit_device(n_reads, reads.n_bases, reads.bases2, reads.seg1_offset, reads.seg1_length, reads.mean_quality, None,
                                 gzip_tallies=Lmax, gzip_output=1)
        gz_batch()
        stream.wait_device()
        t1 = time.perf_counter()
        gz_batch()
        for i in range(k_gz):
            if i + 1 < k_gz:
                gz_batch()
            stream.wait_device()
        gzip_rate = k_gz * n_reads / (time.perf_counter() - t1)

    # what the chip gives the bare probe pattern on this index, measured now (a few ms): the honest ceiling of the probe kernel
    gather_roof = None
    if not rows_mode and not sparse_mode:
        k1_rate = (fetches / (k1_ms / max(k1_n, 1) * 1e-3)) if k1_ms > 0 else 0.0
        roof = {pol: index.gather_roof(nt=(pol == "nt")) for pol in ("default", "nt")}
        best = max(roof.values())
        gather_roof = {"default_policy_fetches_per_s": roof["default"], "nt_policy_fetches_per_s": roof["nt"],
                       "kernel_frac_of_roof": (k1_rate / best) if best > 0 else None,
                       "pattern": "independent uniformly random fetches of 8*W bytes, one per thread in flight, 32 wavefronts per CU"}
    out = None
    if rank == 0:
        k1_avg = k1_ms / max(k1_n, 1)
        traffic = pmc_traffic(args.workload, n_reads, L if Lmax == L else -1)
        achieved = alg_bytes / (k1_avg * 1e-3) / 1e9 if k1_avg > 0 else 0.0
        out = {
            "metric": "classified reads/sec + achieved HBM GB/s vs roofline, 5 kb reads, 39 GB index",
            "value": (1 if rows_mode else world) * args.steps * n_reads / elapsed,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if rows_mode else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": wl["desc"],
                       "input": "device-resident packed reads (2-bit bases + per-read mean-quality / gzip-ratio columns already in HBM when "
                                "the timed region starts); host FASTQ parsing, mean quality and the gzip column are excluded",
                       "timed_region": "length ordering -> minimise+probe -> count -> KDE model + call, %d distinct read set(s) cycled, "
                                       "%d batches in flight" % (n_sets, max(1, min(args.inflight, 3))),
                       "ms_per_step_without_kernel_events": ms_no_events,
                       "reads_per_step_per_gpu": n_reads, "read_len": L, "read_len_max": Lmax, "bases_per_step_per_gpu": int(last_reads.n_bases),
                       "index_bytes": S * ((B + 63) // 64) * 8, "ranks_seen": ranks_seen,
                       "sharding": ("index rows sharded over ranks, same batch on every rank, one sum all-reduce of probe words per batch" if rows_mode else
                                    "index rows AND reads sharded over ranks; per batch one all-to-all of 4-byte row queries and one of the row words back" if sparse_mode else
                                    "reads sharded over ranks, full index replica per GPU, no data-path collective"),
                       "mean_minimisers_per_read": total_min / n_reads, "borderline_reads": int(flags.sum()), "row_log_reruns": int(reruns),
                       "summary_counts": dict([(categories[c], int(summary[c])) for c in range(ncat)] + [("unclassified", int(summary[ncat]))]),
                       "setup_seconds": round(setup_s, 1), "pcie_inclusive_reads_per_s": pcie_rate,
                       "with_gzip_column_on_device_reads_per_s": gzip_rate,
                       "pcie_leg": None if pcie_rate is None else "20 batches through chn_batch_submit with pinned host buffers, three in flight; fill and drain inside the timed leg"},
            "roofline": {"bound": "hbm", "kernel": "k_minimise_probe", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic["traffic_bytes"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": k1_avg,
                         "gathers_per_s": (fetches / (k1_avg * 1e-3)) if k1_avg > 0 else 0.0,
                         "gathers_issued_per_launch": int(fetches), "gathers_nominal_per_launch": int(total_min * index.desc.hash_funs),
                         "gather_roof": gather_roof,
                         "note": "HBM-bound on random row probes: each probe uses 8*W bytes of a 128-byte line, so traffic/algorithmic ~ 8x is "
                                 "line granularity, not re-reads; gather_roof is the rate THIS device gave, in this run, for nothing but uniformly "
                                 "random row fetches from this index (chn_index_gather_roof), and the kernel's probe rate as a fraction of it",
                         "other_kernels_avg_ms": {"k_count_wavelog": k2_ms / max(k2_n, 1), "k_model_call": k3_ms / max(k3_n, 1),
                                                  "whole_chain": chain_ms / max(chain_n, 1),
                                                  "note": "count and model+call run on a side stream under the next batch's minimise+probe"}},
        }
        if rehearsal:
            out["rehearsal"] = True  # ranks share GPUs / gloo backend: plumbing check only, never a scaling figure
        if world == 1 and not args.no_cpu_baseline and not rows_mode and not sparse_mode and not paired:
            out["cpu_baseline"] = cpu_baseline(api, index, last_reads, res, n_reads, args, local, categories, b2c)
    stream.destroy()
    if stream_b is not None:
        stream_b.destroy()
    index.destroy()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def launch_ranks(args):
    """`python bench.py --gpus N` from a plain shell: start the N ranks as children (torch.distributed.run, one process per
    GPU) BEFORE this process touches the GPU, wait for them and relay rank 0's JSON line."""
    import socket
    import subprocess
    if args.backend == "nccl" and not args.launch_check:
        import torch
        ndev = torch.cuda.device_count()  # does not initialise the GPU in this (parent) process
        if args.gpus > ndev:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible; RCCL needs one GPU per rank "
                             "(--backend gloo rehearses the launch on fewer GPUs)" % (args.gpus, ndev))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{"):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line:
        print(line, flush=True)
    if rc != 0 or not line:
        raise SystemExit(rc or 1)


def launch_check(args, rank, world, use_dist):
    """everything bench.py does around the GPU work -- rendezvous, barrier, timed region, MAX over ranks, one JSON line from rank
    0 -- with a sleep standing in for the steps.  Runs without a GPU (gloo): the CPU test of the --gpus N launch path."""
    import torch
    dist = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo" if args.backend == "gloo" or torch.cuda.device_count() < world else "nccl")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1) * args.steps)
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ranks_seen = 1
    if dist:
        t = torch.tensor([elapsed, 1.0], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, ranks_seen = float(tmax[0].item()), int(round(float(t[1].item())))
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "rehearsal": True, "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps,
                          "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "value": None}), flush=True)


def pmc_traffic(workload, n_reads, read_len):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes of this same command (FETCH_SIZE and
    WRITE_SIZE in separate passes, gfx950 correction applied; see profiles/rNN/pmc_traffic.json).  bench.py cannot run
    the profiler on itself, so the most recent committed measurement for the same workload shape is reported, stamped
    with the commit and date it was taken at so that a stale figure is visible."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")), reverse=True):
        try:
            doc = json.load(open(f))
        except (OSError, ValueError):
            continue
        d = doc.get(workload)
        if d and d.get("reads_per_launch") == n_reads and d.get("read_len") == read_len:
            return {"traffic_bytes": d["traffic_bytes"],
                    "source": "%s (measured at commit %s, %s)" % (os.path.relpath(f, ROOT), doc.get("commit", "unrecorded"), doc.get("date", "undated"))}
    return None


def cpu_baseline(api, index, reads, res, n_reads, args, device, categories, b2c):
    """The CPU oracle (a port: restatement of the reference path with plain-word probes, no per-read gzip) timed on a
    bounded sample of the same reads against the same index, on this box's host cores; also re-checks GPU parity on
    that sample."""
    from charon_amd import pack
    from oracle import pyoracle as po
    d = index.desc
    oidx = po.Index.new(d.bins, d.bin_size, b2c, categories, k=d.kmer_size, w=d.window_size, nhash=d.hash_funs)
    words = oidx.words()
    chunk_rows = max(1, (1 << 30) // (8 * d.bin_words))
    for r0 in range(0, d.bin_size, chunk_rows):
        nr = min(chunk_rows, d.bin_size - r0)
        rc = api.lib().chn_index_download_rows(index.h, r0, nr, words[r0 * d.bin_words:].ctypes.data)
        if rc != 0:
            raise RuntimeError("index download failed")
    # Threads: what this job may really use -- min(cgroup CPU quota, affinity mask), capped at the reference's uint8 -t (255;
    # src/dehost_main.cpp:341 num_threads(opt.threads)).  Reported field by field so that a reader can tell a 256-thread host from a 16-core share.
    host = host_cpu_info()
    threads = host["cores"]
    sample = args.cpu_sample_reads or 32768
    sample = min(sample, n_reads)
    lens = api.device_download(device, reads.seg1_length, sample * 4, np.uint32)
    offs = api.device_download(device, reads.seg1_offset, sample * 8, np.uint64)
    nb = int(offs[-1] + ((int(lens[-1]) + 63) // 64) * 64)
    seqs = pack.unpack_reads(api.device_download(device, reads.bases2, nb // 4, np.uint32), offs, lens)
    cat, o = b"".join(seqs), np.concatenate([[0], np.cumsum(lens.astype(np.uint64))]).astype(np.uint64)
    # calibrate on 256 reads, size the timed sample to ~3 s (at most the sample), then time it FIVE times: median + spread
    r = oidx.process_reads(cat[:int(o[256])] if sample > 256 else cat, o[:257] if sample > 256 else o, threads=threads)
    rate = min(256, sample) / max(r["seconds"], 1e-6)
    n_timed = int(min(sample, max(256, rate * 3)))
    runs = []
    for _ in range(5):
        r = oidx.process_reads(cat[:int(o[n_timed])], o[:n_timed + 1], threads=threads)
        runs.append(n_timed / r["seconds"])
    runs.sort()
    gpu_call = api.device_download(device, res.call, n_reads, np.uint8)[:n_timed]
    gpu_nh = api.device_download(device, res.num_hashes, n_reads * 4, np.uint32)[:n_timed]
    gpu_cnt = api.device_download(device, res.counts, n_reads * 8, np.uint32).reshape(-1, 2)[:n_timed]
    gpu_unq = api.device_download(device, res.unique_counts, n_reads * 8, np.uint32).reshape(-1, 2)[:n_timed]
    parity = bool(np.array_equal(gpu_call, r["call"]) and np.array_equal(gpu_nh, r["num_hashes"]) and
                  np.array_equal(gpu_cnt, r["counts"]) and np.array_equal(gpu_unq, r["unique"]))
    # the reference's loop also gzips every read for its `compression` column (src/utils.cpp:114-124): same port with that
    # column switched on, on a smaller sample (reported beside the hot-path-only figure, not instead of it)
    n_gz = min(n_timed, 8192)
    thr = po.default_thresholds(with_gzip=True)
    rg = oidx.process_reads(cat[:int(o[n_gz])], o[:n_gz + 1], threads=threads, thr=thr)
    # reference-faithful form: Elias-Fano get_int probes as the reference keeps the IBF compressed, per-read gzip,
    # at -t 1 and -t N.  The oracle builds its sd_vector straight from the downloaded plain words, in parallel (39 GB plain ->
    # ~35 GB compressed, next to the plain copy in host memory).
    faithful = None
    # building the sd_vector of a 39 GB index takes ~40 s on a 16-core share; with a smaller one it would dominate the run
    if d.bin_size * d.bin_words * 8 <= (2 << 30) or (d.bin_size * d.bin_words * 8 <= (64 << 30) and threads >= 16):
        t0 = time.perf_counter()
        oidx.compress()
        oidx.use_ef(True)
        t_build = time.perf_counter() - t0
        n1 = min(n_timed, 2048)   # (the -t 1 figure from at least 2 000 reads)
        r1 = oidx.process_reads(cat[:int(o[n1])], o[:n1 + 1], threads=1, thr=thr)
        nN = min(n_timed, max(2048, int(n1 / max(r1["seconds"], 1e-6) * threads * 3)))
        rN = oidx.process_reads(cat[:int(o[nN])], o[:nN + 1], threads=threads, thr=thr)
        ok = bool(np.array_equal(gpu_call[:nN], rN["call"]) and np.array_equal(gpu_nh[:nN], rN["num_hashes"]) and
                  np.array_equal(gpu_cnt[:nN], rN["counts"]) and np.array_equal(gpu_unq[:nN], rN["unique"]))
        faithful = {"t1_reads_per_s": n1 / r1["seconds"], "tN_reads_per_s": nN / rN["seconds"], "threads": threads,
                    "sample": "%d reads at -t 1, %d at -t %d; sd_vector (Elias-Fano) probes + per-read gzip column, the form the "
                              "reference's loop has (restatement of the reference CPU path, not the reference)" % (n1, nN, threads),
                    "ef_build_seconds": round(t_build, 1), "gpu_parity_on_sample": ok}
    oidx.free()
    return {"value": runs[len(runs) // 2], "unit": "reads/s", "cores": threads, "kind": "port",
            "threads": threads, "cpus_allowed": host["cpus_allowed"], "cgroup_cpu_quota": host["cgroup_cpu_quota"], "cpu_model": host["cpu_model"],
            "host_logical_cpus": host["logical_cpus"],
            "runs": len(runs), "value_min": runs[0], "value_max": runs[-1], "spread": (runs[-1] - runs[0]) / runs[len(runs) // 2],
            "value_with_gzip_column": n_gz / rg["seconds"], "reference_faithful_form": faithful,
            "sample": "%d of the same reads vs the same index (downloaded from HBM), timed %d times (value = median), oracle hot path only: "
                      "minimisers + plain-word IBF probes + counts + KDE + call, OpenMP over reads on %d threads, no per-read gzip column"
                      % (n_timed, len(runs), threads),
            "gpu_parity_on_sample": parity}


def host_cpu_info():
    """what the CPU side of this job may use: affinity mask, cgroup CPU quota (v2 cpu.max, v1 cfs quota), CPU model; cores = the smaller of
    the first two, capped at 255 (the reference's -t is a uint8)"""
    allowed = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    cores = allowed if quota is None else max(1, min(allowed, int(quota + 0.5)))
    return {"cpus_allowed": allowed, "cgroup_cpu_quota": quota, "cpu_model": model, "logical_cpus": os.cpu_count(), "cores": max(1, min(cores, 255))}


if __name__ == "__main__":
    main()
