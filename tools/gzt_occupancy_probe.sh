#!/bin/bash
# k_gzip_tally: how does its time go with the number of wavefronts per CU?  (LDS padding lowers the occupancy; gpurun)
ROOT=$GRAFT_REPO_ROOT
W=/tmp/clip
[ -f $W/reads.fastq ] || python3 $ROOT/tools/cli_throughput.py 200000 $W --gen-only > /dev/null 2>&1
cd /tmp
for pad in 0 3000 8000 16000 40000; do
  rm -rf /tmp/p_occ
  CHN_GZT_LDS_PAD=$pad rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_occ -- $ROOT/charon_amd/bin/charon dehost --db $W/bench.idx -t 16 --log $W/c.log $W/reads.fastq > $W/o.tsv 2> /dev/null
  f=$(find /tmp/p_occ -name '*kernel_stats.csv' | head -1)
  echo "pad $pad: $(grep k_gzip_tally $f | cut -d, -f2-4)"
done
