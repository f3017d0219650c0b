#!/bin/bash
# exercise bench.py's multi-rank control flow with two ranks on ONE GPU (gloo control plane; RCCL refuses two ranks per
# device, so only the modes without a data-path collective can run this way): config 5 and config 4 --exchange replicate
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export BENCH_DIST_BACKEND=gloo
for args in "--config 5" "--config 4 --exchange replicate"; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 \
    bench.py --gpus 2 --steps 20 --warmup 5 --min-seconds 0.1 --no-cpu-baseline --crt-channels 2000000 --crt-blocks 20 $args 2>&1 | tail -1 | cut -c1-700
done
