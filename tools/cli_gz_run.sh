#!/bin/bash
# CLI throughput on compressed input: plain text vs one-stream .gz vs BGZF vs .bz2 (gpurun; writes gpurun_out/cli_gz.txt)
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
# .bz2 (this build's own block-parallel decoder): the first 20 000 reads (python's bz2 compresses 12 MB/s), against the same reads as plain text
NB=20000
head -n $((NB * 4)) $W/reads.fastq > $W/part.fastq
python3 -c "import bz2,sys; open(sys.argv[2],'wb').write(bz2.compress(open(sys.argv[1],'rb').read(), 9))" $W/part.fastq $W/part.fastq.bz2
ls -l $W/part.fastq $W/part.fastq.bz2 >> $OUT
for f in part.fastq part.fastq.bz2; do
  for t in 1 16; do
    s=$(date +%s.%N)
    CHARON_TIMING=1 $EXE dehost --db $W/bench.idx -t $t --log $W/c.log $W/$f > $W/out_$f.tsv 2> $W/err.txt
    e=$(date +%s.%N)
    python3 -c "print('$f -t $t: %.3f s -> %.0f reads/s' % ($e - $s, $NB / ($e - $s)))" >> $OUT
    grep "reader thread" $W/err.txt >> $OUT
  done
done
cmp $W/out_part.fastq.tsv $W/out_part.fastq.bz2.tsv && echo "TSV identical for .bz2 and plain text" >> $OUT
cmp $W/out_reads.fastq.tsv $W/out_plain.fastq.gz.tsv && cmp $W/out_reads.fastq.tsv $W/out_blocked.fastq.gz.tsv && echo "TSV identical for the three inputs" >> $OUT
cat $OUT
