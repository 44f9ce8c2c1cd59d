#!/bin/bash
# SQ counters of k_minimise_probe in a bench workload (gpurun): is a wavefront's time spent issuing or waiting?
#   WORKLOAD=39g|cfg2|cfg5 (default 39g)   KERNEL='k_minimise_probe<2, 1, 23, false>' (substring of the kernel name)   CHARON_HIP_LIB=<diag build>
ROOT=$GRAFT_REPO_ROOT
WORKLOAD=${WORKLOAD:-39g}
KERNEL=${KERNEL:-k_minimise_probe<2, 1, 23, false>}
export TMPDIR=/tmp KERNEL
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; do
  rm -rf /tmp/p_k1
  rocprofv3 --pmc $set --output-format csv -d /tmp/p_k1 -- python3 $ROOT/bench.py --workload $WORKLOAD --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > /dev/null 2>&1
  f=$(find /tmp/p_k1 -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, os, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if os.environ["KERNEL"] in r["Kernel_Name"]:
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print("%-24s launches %3d  avg %.4g" % (k, n, v / n))
PY
done
