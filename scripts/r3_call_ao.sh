#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for m in fm fmvar; do
  timeout 300 python scripts/scale_kernels_probe.py 1.5 $m 2>&1 | tail -1 | sed "s/^{/{\"path\": \"demod_fm_lanes, launch bounds (64,1)\", /" | tee -a gpurun_out/r3_fm_lanes.jsonl
done
