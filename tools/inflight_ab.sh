# A/B of two vs three device batches in flight in bench.py's timed loop (1 x MI355X); writes gpurun_out/r3a/*
set -e
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for d in 2 3; do
  python bench.py --no-cpu-baseline --no-pcie --inflight $d --steps 8 > $O/def_$d.json 2> $O/def_$d.err
  python bench.py --no-cpu-baseline --no-pcie --inflight $d --read-len 500 --read-len-max 50000 --steps 6 > $O/mixed_$d.json 2> $O/mixed_$d.err
  python bench.py --no-cpu-baseline --no-pcie --inflight $d --workload cfg5 --steps 8 > $O/cfg5_$d.json 2> $O/cfg5_$d.err
  python bench.py --no-cpu-baseline --no-pcie --inflight $d --workload cfg2 --steps 8 > $O/cfg2_$d.json 2> $O/cfg2_$d.err
done
for f in $O/*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r=d["roofline"]
print("%-12s value=%.4g %s  ms/step=%.2f  K1=%.2f ms  frac=%.4f gathers=%.3g" % (sys.argv[1].split("/")[-1][:-5], d["value"], d["unit"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r.get("gathers_per_s",0)))
PY
done
