#!/bin/bash
# A/B of diagnostics builds on one box: tools/diag_ab.sh <outdir> <variant>...   ("normal" = the shipped library)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
B="python3 bench.py --workload 39g --steps 5 --warmup 2 --no-cpu-baseline --no-pcie"
./tools/gather_alloc_bench 39 0 1 | tee $OUT/gather.txt
for v in "$@"; do
  if [ $v = normal ]; then $B > $OUT/bench_$v.json 2>$OUT/bench_$v.err; else CHARON_HIP_LIB=$PWD/tools/diag/libcharon_hip_$v.so $B > $OUT/bench_$v.json 2>$OUT/bench_$v.err; fi
  python3 -c "
import json,sys
try:
    d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1]); print('%-22s step %.2f ms  K1 %.2f ms  %.2f Ggather/s' % ('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['gathers_per_s']/1e9))
except Exception as e: print('$v', 'failed', e)
"
done
