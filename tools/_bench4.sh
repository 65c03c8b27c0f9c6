timeout 900 python bench.py --steps 5 --warmup 2 --batch 256 --bands 50 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_houston.json
timeout 900 python bench.py --steps 3 --warmup 1 --batch 64 --precision fp32 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_fp32.json
timeout 900 python bench.py --steps 5 --warmup 2 --batch 64 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_b64.json
timeout 900 python bench.py --steps 5 --warmup 2 --batch 1024 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_b1024.json
python - <<PY
import json
for n in ("houston","fp32","b64","b1024"):
    d=json.loads(open(f"gpurun_out/bench_{n}.json").read())
    print(n, d["value"], "samples/s", d["ms_per_step"], "ms/step frac", d["step_mfma_frac"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
