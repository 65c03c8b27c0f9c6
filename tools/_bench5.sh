for extra in "" "--no-profile"; do
timeout 900 python bench.py --steps 10 --warmup 3 --batch 256 --no-cpu-baseline $extra 2>&1 | tail -1 > /tmp/b.json
python - <<PY
import json
d=json.loads(open("/tmp/b.json").read())
ks=d.get("kernels") or {}
print("[$extra]", d["value"], "samples/s", d["ms_per_step"], "ms/step; kernel sum per step", round(sum(v["total_ms"] for v in ks.values())/d["steps"],2))
for k,v in ks.items(): print("   %-18s avg %9.1f us x %4d  tot/step %.2f ms" % (k, v["avg_us"], v["launches"], v["total_ms"]/d["steps"]))
PY
done
