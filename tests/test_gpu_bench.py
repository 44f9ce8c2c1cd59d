"""bench.py's contract, on the toy workload: one JSON line with the fields the driver reads, a roofline object for the dominant kernel
and a cpu_baseline object whose sample was checked against the GPU rows -- so that a change to the bench cannot silently drop a field."""
import json
import os
import subprocess
import sys

import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _bench(*args, base=("--workload", "small", "--steps", "3", "--warmup", "1")):
    p = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py")] + list(base) + list(args),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE JSON line
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = _bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "reads/s" and d["dtype"] == "u64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - d["config"]["reads_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert "workload" in d["config"] and "input" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["avg_launch_ms"] <= d["ms_per_step"] * 1.05  # the kernel cannot take longer than the step it is part of
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["gpu_parity_on_sample"] is True
    # a stated, repeatable baseline: what the job may use (affinity, cgroup quota), the threads it did use, several timed runs
    for k in ("threads", "cpus_allowed", "cgroup_cpu_quota", "cpu_model", "runs", "value_min", "value_max", "spread"):
        assert k in c, k
    assert c["threads"] == c["cores"] <= c["cpus_allowed"] and c["runs"] >= 3 and c["value_min"] <= c["value"] <= c["value_max"]
    # the probe kernel reports the row fetches it issued: all h per minimiser, or fewer for this 2-bin index (one row at a time while the AND lives)
    assert 0 < r["gathers_issued_per_launch"] < r["gathers_nominal_per_launch"]
    # informational legs beside `value`: the same batches through host buffers, and with the gzip column computed on the device as well
    assert d["config"]["pcie_inclusive_reads_per_s"] > 0 and 0 < d["config"]["with_gzip_column_on_device_reads_per_s"] <= d["value"] * 1.05
    # sub-millisecond steps are also timed without kernel events, and that figure cannot exceed the one with them by much
    ne = d["config"]["ms_per_step_without_kernel_events"]
    assert ne is None or 0 < ne <= d["ms_per_step"] * 1.25


def test_bench_row_sharded_modes_run_at_one_rank():
    for mode in ("rows", "rows-dense"):
        d = _bench("--shard", mode, "--no-cpu-baseline")
        assert d["value"] > 0 and d["n_gpus"] == 1
        assert "sharded" in d["config"]["sharding"]


def test_bench_small_paired_batches_report_the_step_without_kernel_events():
    """BASELINE configs[4] at the CLI's batch size: paired 2 x 150 b reads, 8 192 pairs per batch, call_category over 8 categories"""
    d = _bench(base=("--workload", "cfg5", "--reads-per-step", "8192", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-pcie"))
    assert d["config"]["reads_per_step_per_gpu"] == 8192 and d["unit"] in ("pairs/s", "reads/s")
    ne = d["config"]["ms_per_step_without_kernel_events"]
    assert ne is not None and 0 < ne <= d["ms_per_step"] * 1.25
    assert d["roofline"]["avg_launch_ms"] <= ne * 1.05  # launches collapsed: the step is the probe kernel plus a few per cent


def test_two_ranks_drive_the_sharded_hip_paths_under_a_real_collective():
    """The N > 1 code paths with the DEVICE in them: bench.py --gpus 2 starts two ranks (torch.distributed.run, gloo,
    both on this box's one GPU -- a rehearsal, labelled so, never a scaling figure).  reads: replicas + the summary all-reduce; rows: the
    sparse exchange (chn_shardx_* under two real all_to_all_v's: queries out, row words back); rows-dense: one sum all-reduce of probe
    words.  Each must classify the same global reads as ONE rank does and arrive at the same summary counters (include/result.hpp:18-25)."""
    common = ("--workload", "small", "--steps", "2", "--warmup", "1", "--read-sets", "1", "--no-cpu-baseline", "--no-pcie")
    n = 4096
    one_2n = _bench("--reads-per-step", str(2 * n), base=common)["config"]["summary_counts"]   # global reads [0, 2n)
    one_n = _bench("--reads-per-step", str(n), base=common)["config"]["summary_counts"]        # global reads [0, n)
    assert sum(one_2n.values()) == 2 * n and sum(one_n.values()) == n and min(one_n.values()) > 0
    for mode, want in (("reads", one_2n), ("rows", one_2n), ("rows-dense", one_n)):
        d = _bench("--gpus", "2", "--backend", "gloo", "--shard", mode, "--reads-per-step", str(n), base=common)
        assert d.get("rehearsal") is True and d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2, (mode, d)
        assert d["config"]["summary_counts"] == want, (mode, d["config"]["summary_counts"], want)
        assert d["scaling"] == ("strong" if mode == "rows-dense" else "weak")
        assert d["value"] > 0 and d["config"]["row_log_reruns"] == 0
