# same-box A/B of the row-gather cache policy (CHN_NT_PROBES overrides the size rule)
for rep in 1 2 3 4; do for nt in 1 0; do CHN_NT_PROBES=$nt python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('39g nt=$nt', round(d['value']/1e6,2), round(d['roofline']['avg_launch_ms'],2))"; done; done
