#!/bin/bash
# round 3, call y: the drop-in with the front end 2 / 3 blocks ahead of the slowest channel (the filter keeps ND = 4 blocks)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for a in 2 3; do
  timeout 300 python scripts/dropin_rate.py 500 1024 HARNESS_AHEAD=$a 2>&1 | tail -1 | tee -a gpurun_out/r3_dropin_ahead.jsonl
  timeout 300 python scripts/dropin_rate.py 500 1024 HARNESS_AHEAD=$a KA9Q_HIP_FDOMAIN=0 KA9Q_HIP_NOISE_SAMPRATE=129600000.0 2>&1 | tail -1 | tee -a gpurun_out/r3_dropin_ahead.jsonl
  timeout 300 python scripts/dropin_rate.py 500 2000 HARNESS_AHEAD=$a KA9Q_HIP_FDOMAIN=0 KA9Q_HIP_NOISE_SAMPRATE=129600000.0 2>&1 | tail -1 | tee -a gpurun_out/r3_dropin_ahead.jsonl
done
