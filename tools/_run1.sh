mkdir -p gpurun_out
./tools/probe_tr > gpurun_out/probe_tr.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/fwd_test.txt
cat gpurun_out/fwd_test.txt
