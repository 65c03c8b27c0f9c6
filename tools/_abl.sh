for d in 0 1 2 4 7; do
MSST_DBG=$d timeout 600 python bench.py --steps 3 --warmup 1 --batch 256 --no-cpu-baseline 2>&1 | tail -1 > /tmp/b.json
python - <<PY
import json
d=json.loads(open("/tmp/b.json").read())
print("dbg=$d block_fwd avg us:", d["kernels"]["block_fwd"]["avg_us"])
PY
done
