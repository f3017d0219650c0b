#!/bin/bash
# round 3, call ab: the linear demodulator at one channel per lane -- parity on the device, cost per channel against the wavefront kernel
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_golden.py tests/test_gpu_parity.py -m gpu -q -x -k "demod or fm_ or coherent or linear or golden or filter2 or partial_rerun" -p no:cacheprovider 2>&1 | tail -4
for w in 1 0; do
  for m in linear pll; do
    CHZ_DEMOD_WAVE=$w timeout 300 python scripts/scale_kernels_probe.py 1.5 $m 2>&1 | tail -1 | sed "s/^{/{\"CHZ_DEMOD_WAVE\": $w, /" | tee -a gpurun_out/r3_demod_lanes.jsonl
  done
done
