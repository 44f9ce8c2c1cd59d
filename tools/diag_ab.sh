#!/bin/bash
# A/B of library builds on ONE box, interleaved: tools/diag_ab.sh <outdir> <rounds> <variant>...   ("normal" = the shipped library,
# anything else = tools/diag/libcharon_hip_<variant>.so from tools/build_diag.sh).  Prints the median K1 time per variant.
OUT=gpurun_out/$1; shift
R=$1; shift
WL=${WORKLOAD:-39g}
mkdir -p $OUT
B="python3 bench.py --workload $WL --steps 8 --warmup 2 --no-cpu-baseline --no-pcie"
./tools/gather_alloc_bench 39 0 1 | tee $OUT/gather.txt
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ $v = normal ]; then $B > $OUT/bench_${v}_$r.json 2>$OUT/bench_$v.err; else CHARON_HIP_LIB=$PWD/tools/diag/libcharon_hip_$v.so $B > $OUT/bench_${v}_$r.json 2>$OUT/bench_$v.err; fi
  done
done
python3 - $OUT "$@" <<'PY'
import json, sys, glob, statistics
out = sys.argv[1]
for v in sys.argv[2:]:
    k1, st, g = [], [], []
    for f in sorted(glob.glob("%s/bench_%s_*.json" % (out, v))):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            k1.append(d["roofline"]["avg_launch_ms"]); st.append(d["ms_per_step"]); g.append(d["roofline"]["gathers_per_s"] / 1e9)
        except Exception as e:
            print(v, "failed", f, e)
    if k1:
        print("%-24s K1 median %.2f ms (%s)  step median %.2f ms  %.2f Ggather/s" % (v, statistics.median(k1), " ".join("%.2f" % x for x in k1), statistics.median(st), statistics.median(g)))
PY
