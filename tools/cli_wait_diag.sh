#!/bin/bash
# per-batch split of chn_batch_wait inside one CLI run (gpurun)
set -e
N=${1:-400000}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/cli_phase
mkdir -p $OUT
W=/tmp/clip
python3 $ROOT/tools/cli_throughput.py $N $W --gen-only > $OUT/gen.log 2>&1
EXE=$ROOT/charon_amd/bin/charon
CHARON_TIMING=1 CHN_DIAG_WAIT=1 $EXE dehost --db $W/bench.idx -t 16 --log $W/c.log $W/reads.fastq > $W/out.tsv 2> $OUT/wait_diag.txt
cat $OUT/wait_diag.txt
