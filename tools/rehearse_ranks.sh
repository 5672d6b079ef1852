#!/bin/bash
# bench.py's multi-rank path with more ranks than GPUs (ranks share the box's one device, synchronise over gloo; RCCL refuses two
# ranks on a device).  Four ranks: a GPU box allows six processes on its card and the launcher counts (six ranks were killed by
# the box's process guard: "7 processes had the GPU open").  The sharded generation, the barrier + max-over-ranks
# timing, the rank report (ranks_seen) and rank 0's line are the code the driver's 1 / 2 / 4 / 8-GPU scaling step runs.
N=${1:-4}
cd "$(dirname "$0")/.."
LFD_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus $N --frames-per-gpu 96 --steps 5 --warmup 1 --gen-workers 2
