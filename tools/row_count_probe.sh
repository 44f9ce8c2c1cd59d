#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
W=/tmp/clir
python3 $ROOT/tools/cli_throughput.py ${1:-60000} $W --gen-only > /dev/null 2>&1
$ROOT/charon_amd/bin/charon dehost --db $W/bench.idx -t 8 --log $W/c.log $W/reads.fastq > $W/out.tsv
wc -l $W/out.tsv; grep -c "^@r" $W/reads.fastq
cut -f2 $W/out.tsv | sort > $W/ids_out.txt; grep "^@r" $W/reads.fastq | cut -c2- | sort > $W/ids_in.txt
comm -3 $W/ids_in.txt $W/ids_out.txt | head; head -c 300 $W/out.tsv; echo; tail -c 200 $W/out.tsv | od -c | tail -5
tail -3 $W/c.log
