mkdir -p gpurun_out
timeout 600 python tools/diag_bwd.py > gpurun_out/diag_bwd.txt 2>&1
tail -80 gpurun_out/diag_bwd.txt
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_forward.py -m gpu -q 2>&1 | tail -30 > gpurun_out/bwd_test.txt
cat gpurun_out/bwd_test.txt
