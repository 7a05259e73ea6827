#!/bin/bash
# two_rank.sh -- the N > 1 bench path on a one-GPU box: two ranks share device 0 (LF_BENCH_BACKEND=gloo test hook)
export LF_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 2 --steps 2 --warmup 1 --genome-mbp 100 --reads 5000 --no-cpu-baseline
