# every bench line quoted in DESIGN.md section 5 (1 x MI355X); writes gpurun_out/final/*.json
set -e
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/default_39g.json
python bench.py --workload cfg2 --no-cpu-baseline > gpurun_out/final/cfg2.json
python bench.py --workload cfg5 --no-cpu-baseline > gpurun_out/final/cfg5.json
python bench.py --workload 39g --read-len 500 --read-len-max 50000 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/final/mixed.json
python bench.py --workload 39g --read-len 300 --reads-per-step 8388608 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/final/short300.json
python bench.py --workload 39g --shard rows --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/rows.json
python bench.py --workload 39g --pcie --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/final/pcie.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/final/torchrun_n1.json
for f in gpurun_out/final/*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r=d["roofline"]; c=d["config"]
print("%-18s value=%.4g %s  ms/step=%.2f  K1=%.2f ms  frac=%.4f  others=%s  pcie=%s  cpu=%s" % (sys.argv[1].split("/")[-1][:-5], d["value"], d["unit"], d["ms_per_step"], r["avg_launch_ms"], r["frac"],
      {k: round(v,2) for k,v in r["other_kernels_avg_ms"].items()}, c.get("pcie_inclusive_reads_per_s"), d.get("cpu_baseline",{}).get("value")))
PY
done
