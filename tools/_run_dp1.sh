export MSST_FORCE_DP=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-600
