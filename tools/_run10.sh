timeout 300 python pretrain.py --depth 2 --batch-size 32 --tiles 64 --epochs 2 --max-steps 4 2>&1 | tail -4
timeout 300 python finetune.py enmap --steps 20 --batch-size 4 2>&1 | tail -3
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 pretrain.py --depth 2 --batch-size 32 --tiles 64 --epochs 1 2>&1 | tail -3
