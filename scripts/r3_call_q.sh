#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r03_bench.err | tail -1 > gpurun_out/r03_bench.json
timeout 1500 bash scripts/gpu_profile.sh r03 2>&1 | tail -12
du -sh gpurun_out
