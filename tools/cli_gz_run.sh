#!/bin/bash
# CLI throughput on compressed input: plain text vs one-stream .gz vs BGZF (gpurun; writes gpurun_out/cli_gz.txt)
set -e
N=${1:-100000}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/cli_gz.txt
W=/tmp/clig
python3 $ROOT/tools/cli_throughput.py $N $W --gen-only > /dev/null 2>&1
gzip -1 -c $W/reads.fastq > $W/plain.fastq.gz
python3 $ROOT/tools/make_bgzf.py $W/reads.fastq $W/blocked.fastq.gz 1 16
ls -l $W/*.fastq* > $OUT
EXE=$ROOT/charon_amd/bin/charon
for f in reads.fastq plain.fastq.gz blocked.fastq.gz; do
  for t in 1 16; do
    s=$(date +%s.%N)
    CHARON_TIMING=1 $EXE dehost --db $W/bench.idx -t $t --log $W/c.log $W/$f > $W/out_$f.tsv 2> $W/err.txt
    e=$(date +%s.%N)
    python3 -c "print('$f -t $t: %.3f s -> %.0f reads/s' % ($e - $s, $N / ($e - $s)))" >> $OUT
    grep "main thread\|reader thread" $W/err.txt >> $OUT
  done
done
cmp $W/out_reads.fastq.tsv $W/out_plain.fastq.gz.tsv && cmp $W/out_reads.fastq.tsv $W/out_blocked.fastq.gz.tsv && echo "TSV identical for the three inputs" >> $OUT
cat $OUT
