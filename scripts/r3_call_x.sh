#!/bin/bash
# round 3, call x: the driver's bench command on the final tree, then the round's rocprofv3 profile (kernel trace + PMC passes)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/*.so
( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
tail -3 gpurun_out/r03_bench.err
timeout 2400 bash scripts/gpu_profile.sh r03 > gpurun_out/r03_profile.log 2>&1
tail -5 gpurun_out/r03_profile.log
du -sh gpurun_out
