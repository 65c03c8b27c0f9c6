for d in 0.0 0.1; do
timeout 900 python bench.py --steps 5 --warmup 2 --batch 256 --no-cpu-baseline --profile-all --dropout $d 2>&1 | tail -1 > /tmp/b.json
python - <<PY
import json
d=json.loads(open("/tmp/b.json").read())
print("dropout $d:", d["value"], "samples/s", d["ms_per_step"], "ms/step", {k: round(v["avg_us"]) for k,v in d["kernels"].items() if k.startswith("block")})
PY
done
