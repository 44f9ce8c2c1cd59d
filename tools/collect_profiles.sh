#!/bin/bash
# Regenerates the evidence under profiles/rNN on the GPU box (run through gpurun; writes to gpurun_out/profNN, copy what is to be
# judged into profiles/rNN afterwards):  bash tools/collect_profiles.sh r02 <commit>
set -e
export TMPDIR=/tmp
R=${1:-r03}
COMMIT=${2:-unknown}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp
# 1. per-kernel time (rocprofv3 --kernel-trace --stats) of the bench command, with the bench line the same run printed
for wl in 39g cfg2 cfg5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$wl -- python3 $ROOT/bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-pcie > $OUT/bench_$wl.log 2>&1
  cp $(find /tmp/p_$wl -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$wl.csv
  grep '^{' $OUT/bench_$wl.log | tail -1 > $OUT/bench_under_profiler_$wl.json
done
# 2. HBM traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md: FETCH_SIZE x2 on gfx950)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $ROOT/bench.py --workload 39g --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $OUT/pmc_$c.log 2>&1
  cp $(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1) $OUT/pmc_$c.csv
done
python3 - $OUT $COMMIT <<'PY'
import csv, sys, json, collections, datetime
out, commit = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open("%s/pmc_%s.csv" % (out, c))):
        k = r["Kernel_Name"].split("(")[0]
        a = agg[k][c]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open(out + "/pmc_summary.csv", "w") as f:
    f.write("kernel,counter,launches,avg_value_KB\n")
    for k, cs in agg.items():
        for c, (n, v) in cs.items():
            f.write("%s,%s,%d,%.1f\n" % (k, c, n, v / n))
k1 = [k for k in agg if "k_minimise_probe<2, 1, 23, false" in k or "k_minimise_probe<2, 1, 23>" in k][0]  # the ordinary launch of the W = 2 row-log kernel
fetch_kb, write_kb = agg[k1]["FETCH_SIZE"][1] / agg[k1]["FETCH_SIZE"][0], agg[k1]["WRITE_SIZE"][1] / agg[k1]["WRITE_SIZE"][0]
traffic = fetch_kb * 1024 * 2 + write_kb * 1024   # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes
json.dump({"commit": commit, "date": datetime.date.today().isoformat(),
           "39g": {"reads_per_launch": 1048576, "read_len": 5000, "kernel": k1, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
                   "traffic_bytes": traffic, "note": "FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024, separate --pmc passes of bench.py --workload 39g"}},
          open(out + "/pmc_traffic.json", "w"), indent=1)
print("traffic per launch: %.1f GB (fetch %.1f GB after x2, write %.1f GB)" % (traffic / 1e9, fetch_kb * 2048 / 1e9, write_kb * 1024 / 1e9))
PY
# 3. gather roofs of this box
cd $ROOT
./tools/gather_alloc_bench 39 > $OUT/gather_alloc_microbench.txt 2>&1
./tools/gather_alloc_bench 1.1 >> $OUT/gather_alloc_microbench.txt 2>&1
# 4. small batches (BASELINE config 5: launch collapse): paired 2 x 150 b, 8 192 and 65 536 pairs per batch
for n in 8192 65536; do
  python3 bench.py --workload cfg5 --reads-per-step $n --steps 200 --warmup 20 --no-cpu-baseline --no-pcie > $OUT/bench_cfg5_$n.json 2>/dev/null
done
# 5. other workload shapes + the default line with the CPU baseline
python3 bench.py --workload 39g --steps 6 --warmup 2 --no-cpu-baseline --read-len 500 --read-len-max 50000 > $OUT/bench_mixed.json 2>/dev/null
python3 bench.py --workload 39g --shard rows --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_rows_sparse.json 2>/dev/null
python3 bench.py --workload 39g --shard rows-dense --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_rows_dense.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2>$OUT/bench_default.err
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print("%-34s value %8.2f M/s  step %7.3f ms  K1 %7.3f ms  frac %.4f  %.1f Ggather/s" % (os.path.basename(f), d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["gathers_per_s"] / 1e9))
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
# 5b. two ranks on this one GPU (gloo rehearsal of the N > 1 paths at the 39 GB shape: plumbing + summary counters, never a scaling figure)
# (both ranks share ONE 288 GB card here: 2 x 262 144 reads per step, so that two sets of exchange buffers fit beside the two index halves / replicas)
for mode in reads rows; do
  python3 bench.py --gpus 2 --backend gloo --workload 39g --shard $mode --steps 4 --warmup 1 --read-sets 1 --reads-per-step 262144 --no-cpu-baseline --no-pcie > $OUT/rehearsal_2ranks_$mode.json 2>$OUT/rehearsal_2ranks_$mode.err || true
done
python3 bench.py --workload 39g --steps 4 --warmup 1 --read-sets 1 --reads-per-step 524288 --no-cpu-baseline --no-pcie > $OUT/rehearsal_1rank_2n.json 2>/dev/null || true
# 5c. SQ counters of the probe kernel (issue vs wait)
WORKLOAD=39g KERNEL="k_minimise_probe<2, 1, 23, false" bash $ROOT/tools/k1_sq_pmc.sh > $OUT/sq_39g.txt 2>&1 || true
WORKLOAD=cfg2 KERNEL="k_minimise_probe<1, 0, 23, false" bash $ROOT/tools/k1_sq_pmc.sh > $OUT/sq_cfg2.txt 2>&1 || true
# 6. the CLI: phase timers + per-kernel GPU time of one run, throughput table, compressed inputs
bash $ROOT/tools/cli_phase_run.sh 400000 > /dev/null 2>&1 || true
cp $ROOT/gpurun_out/cli_phase/phase.txt $OUT/cli_phase.txt 2>/dev/null || true
cp $ROOT/gpurun_out/cli_phase/kernel_stats_cli.csv $OUT/kernel_stats_cli.csv 2>/dev/null || true
python3 $ROOT/tools/cli_throughput.py 400000 /tmp/clib > $OUT/cli_throughput_raw.txt 2>&1 || true
rm -rf /tmp/clib /tmp/clip
python3 $ROOT/tools/cli_steady_state.py 4000000 /tmp/css 16 16 1 > $OUT/cli_steady_state.txt 2>&1 || true
rm -rf /tmp/css
bash $ROOT/tools/cli_gz_run.sh 100000 > /dev/null 2>&1 || true
cp $ROOT/gpurun_out/cli_gz.txt $OUT/cli_gz.txt 2>/dev/null || true
./tools/h2d_bench > $OUT/h2d_microbench.txt 2>&1 || true
