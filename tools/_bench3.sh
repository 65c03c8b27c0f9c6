for cfg in "256 16" "512 16" "768 32" "1024 32"; do
set -- $cfg
MSST_BWD_GRID=$1 MSST_TOK_CHUNKS=$2 timeout 900 python bench.py --steps 5 --warmup 2 --batch 256 --no-cpu-baseline 2>&1 | tail -1 > /tmp/b.json
python - <<PY
import json
d=json.loads(open("/tmp/b.json").read())
print("grid $1 tok $2:", d["value"], "samples/s", d["ms_per_step"], "ms/step", {k: round(v["avg_us"]) for k,v in d["kernels"].items() if k.startswith("block") or k.startswith("tokenize_b") or k.startswith("reduce")})
PY
done
