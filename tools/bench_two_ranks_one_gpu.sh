#!/bin/bash
# rehearsal of bench.py's N > 1 code path on a one-GPU box: two ranks share the card, gloo moves the device tensors
# (RCCL needs one GPU per rank).  Numbers mean nothing; the point is that the path runs and rank 0 prints its line.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${PORT:-29555} WORLD_SIZE=2 LOCAL_RANK=0 MMHIP_DIST_BACKEND=gloo
RANK=1 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 16 --no-cpu-baseline "$@" > /tmp/rank1.out 2>&1 &
P1=$!
RANK=0 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 16 --no-cpu-baseline "$@"
R=$?
wait $P1
echo "rank0 rc=$R rank1 rc=$?"; tail -3 /tmp/rank1.out
