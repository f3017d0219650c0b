#!/bin/bash
# Exercise bench.py's multi-rank control flow with TWO ranks on ONE GPU (gloo control plane, replicate mode): every rank
# runs the whole per-GPU workload on device 0, rank 0 prints the aggregated JSON line.  Not a performance measurement.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29544} WORLD_SIZE=2 LOCAL_RANK=0 BENCH_DIST_BACKEND=gloo
RANK=1 python bench.py --gpus 2 --steps 300 --warmup 30 --crt-channels 3000000 > /tmp/rank1.out 2> /tmp/rank1.err &
P1=$!
RANK=0 python bench.py --gpus 2 --steps 300 --warmup 30 --crt-channels 3000000 2> /tmp/rank0.err | tail -1
wait $P1; echo "rank1 exit $?  stdout: $(cat /tmp/rank1.out | head -c 300)"
tail -2 /tmp/rank1.err
