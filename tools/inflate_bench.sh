#!/bin/bash
# own inflate vs zlib on a one-stream .gz, by thread count (gpurun; host-side only)
set -e
N=${1:-100000}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/inflate_bench.txt
W=/tmp/clig
python3 $ROOT/tools/cli_throughput.py $N $W --gen-only > /dev/null 2>&1
gzip -6 -c $W/reads.fastq > $W/plain6.fastq.gz
ls -l $W/reads.fastq $W/plain6.fastq.gz > $OUT
for t in 1 2 4 8 16 32; do
  echo "threads $t" >> $OUT
  CHARON_DIAG_INFLATE=1 CHARON_READER_THREADS=$t $ROOT/charon_amd/bin/charon _inflate $W/plain6.fastq.gz >> $OUT 2> $W/diag.txt
  tail -2 $W/diag.txt >> $OUT
done
cat $OUT
