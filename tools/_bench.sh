mkdir -p gpurun_out
timeout 900 python bench.py --steps 5 --warmup 2 --batch 256 --no-cpu-baseline --profile-all 2>&1 | tail -1 > gpurun_out/bench_cur.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_cur.txt").read())
print(d["value"], "samples/s", d["ms_per_step"], "ms/step", "mfma_frac", d["step_mfma_frac"], d["roofline"])
for k,v in d["kernels"].items(): print("   %-18s avg %9.1f us x %4d  share %.3f" % (k, v["avg_us"], v["launches"], v["share"]))
PY
