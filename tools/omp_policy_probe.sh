#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
W=/tmp/clig
N=${1:-300000}
[ -f $W/plain.fastq.gz ] || { python3 $ROOT/tools/cli_throughput.py $N $W --gen-only > /dev/null 2>&1; gzip -1 -c $W/reads.fastq > $W/plain.fastq.gz; }
EXE=$ROOT/charon_amd/bin/charon
for pol in default passive active; do
  for f in plain.fastq.gz reads.fastq; do
    for rep in 1 2; do
      s=$(date +%s.%N)
      if [ $pol = default ]; then CHARON_TIMING=1 $EXE dehost --db $W/bench.idx -t 16 --log $W/c.log $W/$f > $W/o.tsv 2> $W/err.txt
      else OMP_WAIT_POLICY=$pol CHARON_TIMING=1 $EXE dehost --db $W/bench.idx -t 16 --log $W/c.log $W/$f > $W/o.tsv 2> $W/err.txt; fi
      e=$(date +%s.%N)
      python3 -c "print('$pol $f: %.3f s' % ($e - $s))"
      grep "reader thread\|main thread" $W/err.txt
    done
  done
done
