# kernel timeline of the sparse row-sharded bench at N = 1 (tools/trace_mixed.sh for --shard rows)
set -e
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
TAG=${1:-rows}; shift || true
for kv in "$@"; do export "$kv"; done
OUT=$ROOT/gpurun_out/r3d
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$TAG -- python3 $ROOT/bench.py --no-cpu-baseline --no-pcie --shard rows --steps 4 --warmup 1 > $OUT/bench_$TAG.log 2>&1
cp $(find /tmp/tr_$TAG -name '*kernel_trace.csv' | head -1) $OUT/kernel_trace_$TAG.csv
python3 - $OUT/kernel_trace_$TAG.csv <<'PY' > $OUT/trace_${TAG}_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].split("(")[0][:40]
    if n.startswith("k_synth") or "rocclr" in n or "k_fill" in n or "k_len" in n or "gather_roof" in n: continue
    print("%-40s q=%s start=%10.3f end=%10.3f dur=%8.3f ms" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
echo "== $TAG $@"; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$TAG.log | tail -1; tail -40 $OUT/trace_${TAG}_summary.txt
