#!/bin/bash
# round 3, call ar: the driver's bench command on the final tree
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
tail -3 gpurun_out/r03_bench.err
