for g in 256 512 1024; do
MSST_BWD_GRID=$g timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-all 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print($g, d['value'], {k: round(v['avg_us']) for k,v in d['kernels'].items() if k.startswith('block') or k=='reduce_slabs'})"
done
