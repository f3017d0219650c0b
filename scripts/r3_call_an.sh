#!/bin/bash
# round 3, call an: where demod_fm_lanes overtakes the wavefront-per-channel FM (bank size sweep, default SNR estimator)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for n in 0.036864 0.073728 0.147456 0.294912; do
  for w in 1 0; do
    CHZ_DEMOD_WAVE=$w timeout 300 python scripts/scale_kernels_probe.py $n fmvar 2>&1 | tail -1 | sed "s/^{/{\"CHZ_DEMOD_WAVE\": $w, /" | tee -a gpurun_out/r3_fm_crossover.jsonl
  done
done
