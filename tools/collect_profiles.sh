set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/prof2
for wl in 39g cfg2 cfg5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$wl -- python3 bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof2/bench_$wl.log 2>&1
  f=$(find /tmp/p_$wl -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/prof2/kernel_stats_$wl.csv
  tail -1 gpurun_out/prof2/bench_$wl.log | cut -c1-400
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 bench.py --workload 39g --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof2/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  python3 - $f $c <<'PY' > gpurun_out/prof2/pmc_$c.txt
import csv,sys,collections
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0]; agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
for k,(n,v) in sorted(agg.items(), key=lambda x:-x[1][1]):
    print(f"{sys.argv[2]},{k},{n},{v/n:.1f}")
PY
  head -4 gpurun_out/prof2/pmc_$c.txt
done
