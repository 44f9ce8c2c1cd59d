for pad in 0 2048 4096 7168 11264 0; do CHN_LDS_PAD=$pad python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pad=$pad', round(d['value']/1e6,2), round(d['roofline']['avg_launch_ms'],2))"; done
