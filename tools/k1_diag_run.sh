#!/bin/bash
# K1 diagnostics on the GPU box: clocks (GRBM_GUI_ACTIVE) of k_minimise_probe vs the pure gather micro-benchmark, SQ/TA/TCP counters,
# and the probe pipeline alone (fake-emit diagnostics build).  Output under gpurun_out/$1.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-k1diag}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --workload 39g --steps 3 --warmup 1 --no-cpu-baseline --no-pcie"
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
pmc() {  # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$n -- $B > $OUT/pmc_$n.log 2>&1
  cp $(find /tmp/pmc_$n -name '*counter_collection.csv' | head -1) $OUT/pmc_$n.csv
}
pmc grbm GRBM_GUI_ACTIVE
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_gb -- $GRAFT_REPO_ROOT/tools/gather_alloc_bench 39 0 1 > $OUT/pmc_gather_grbm.log 2>&1
cp $(find /tmp/pmc_gb -name '*counter_collection.csv' | head -1) $OUT/pmc_gather_grbm.csv
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
pmc sq2 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
pmc ta TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
cd $GRAFT_REPO_ROOT
$B > $OUT/bench_normal.json 2> $OUT/bench_normal.err
CHARON_HIP_LIB=$GRAFT_REPO_ROOT/tools/diag/libcharon_hip_fake_emit.so $B > $OUT/bench_fake_emit.json 2> $OUT/bench_fake_emit.err
python3 - $OUT <<'PY'
import csv, sys, glob, os, json, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print("==", os.path.basename(f))
    for k, cs in agg.items():
        if "minimise" in k or "gather" in k or "count_wavelog" in k:
            for c, (n, v, ms) in cs.items():
                print("  %-60s %-28s n=%d avg=%.4g avg_ms=%.3f" % (k, c, n, v / n, ms / n))
for n in ("normal", "fake_emit"):
    try:
        d = json.loads(open(out + "/bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, "ms_per_step", d["ms_per_step"], "k1_ms", d["roofline"]["avg_launch_ms"], "gathers/s", d["roofline"]["gathers_per_s"], "min/read", d["config"]["mean_minimisers_per_read"])
    except Exception as e:
        print(n, "failed", e)
PY
