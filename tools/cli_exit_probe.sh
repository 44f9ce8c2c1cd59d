#!/bin/bash
# where does the time between dehost_main's return and the process's disappearance go?  (gpurun)
N=${1:-400000}
ROOT=$GRAFT_REPO_ROOT
W=/tmp/clip
python3 $ROOT/tools/cli_throughput.py $N $W --gen-only > /dev/null 2>&1
EXE=$ROOT/charon_amd/bin/charon
run() {
  s=$(date +%s.%N)
  env "$@" CHARON_TIMING=1 $EXE dehost --db $W/bench.idx -t 16 --log $W/c.log $W/reads.fastq > $W/out.tsv 2> $W/err.txt
  e=$(date +%s.%N)
  ret=$(grep -o "wall clock [0-9.]*" $W/err.txt | cut -d' ' -f3)
  python3 -c "print('$*: wall %.3f s, after dehost_main returned: %.3f s' % ($e - $s, $e - $ret))"
}
run A=1
run A=1
run CHARON_NO_MMAP=1
run CHARON_NO_MMAP=1
run CHARON_FULL_EXIT=1
