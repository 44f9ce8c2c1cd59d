#!/bin/bash
# Phase timers and per-kernel GPU time of one `charon dehost -t 16` run on a synthetic FASTQ (gpurun; writes gpurun_out/cli_phase)
set -e
export TMPDIR=/tmp
N=${1:-400000}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/cli_phase
mkdir -p $OUT
W=/tmp/clip
python3 $ROOT/tools/cli_throughput.py $N $W --gen-only > $OUT/gen.log 2>&1
EXE=$ROOT/charon_amd/bin/charon
for t in 16 16 1; do
  s=$(date +%s.%N)
  CHARON_TIMING=1 $EXE dehost --db $W/bench.idx -t $t --log $W/c.log $W/reads.fastq > $W/out.tsv 2> $OUT/err_t$t.txt
  e=$(date +%s.%N)
  echo "-t $t: wall $(python3 -c "print('%.3f' % ($e - $s))") s for $N reads (exec at wall clock $s, exit seen at $e)" >> $OUT/phase.txt
  cat $OUT/err_t$t.txt >> $OUT/phase.txt
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_cli -- $EXE dehost --db $W/bench.idx -t 16 --log $W/c.log $W/reads.fastq > $W/out2.tsv 2> $OUT/prof_err.txt
cp $(find /tmp/p_cli -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_cli.csv
cmp $W/out.tsv $W/out2.tsv && echo "TSV identical under the profiler" >> $OUT/phase.txt
cat $OUT/phase.txt
