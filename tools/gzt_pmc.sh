#!/bin/bash
# SQ counters of k_gzip_tally inside one CLI run (gpurun): where do a wavefront's cycles go?
ROOT=$GRAFT_REPO_ROOT
W=/tmp/clip
export TMPDIR=/tmp KERNEL=${KERNEL:-k_gzip_tally}
[ -f $W/reads.fastq ] || python3 $ROOT/tools/cli_throughput.py 100000 $W --gen-only > /dev/null 2>&1
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES"; do
  rm -rf /tmp/p_gzt
  rocprofv3 --pmc $set --output-format csv -d /tmp/p_gzt -- $ROOT/charon_amd/bin/charon dehost --db $W/bench.idx -t 16 --log $W/c.log $W/reads.fastq > $W/o.tsv 2> /dev/null
  f=$(find /tmp/p_gzt -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if __import__("os").environ.get("KERNEL", "k_gzip_tally") in r["Kernel_Name"]:
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print("%-24s launches %3d  avg %.4g" % (k, n, v / n))
PY
done
