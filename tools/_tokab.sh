for c in 16 32 64 128; do
MSST_TOK_CHUNKS=$c timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-all 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print($c, d['value'], {k: round(v['avg_us']) for k,v in d['kernels'].items() if k in ('tokenize_bwd','reduce_slabs','tokenize_fwd','head_bwd','head_fwd','adamw','prep_weights')})"
done
