#!/bin/bash
# round 3, call w: the whole GPU suite on the final tree
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r3_gpu_suite.txt
