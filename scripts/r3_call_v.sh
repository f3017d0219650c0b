#!/bin/bash
# round 3, call v: masters of any length (chirp-z) -- parity on the device, then cost and accuracy at full size
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "outside_the_compiled or forward_matches" -p no:cacheprovider 2>&1 | tail -3
timeout 600 python scripts/r3_blue_master_probe.py 2>&1 | tail -3 | tee gpurun_out/r3_blue_master.jsonl
