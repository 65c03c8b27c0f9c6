mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/gpu_tests.txt
cat gpurun_out/gpu_tests.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 900 python bench.py --steps 5 --warmup 2 --batch 64 2>&1 | tail -5 > gpurun_out/bench_b64.txt
cat gpurun_out/bench_b64.txt
