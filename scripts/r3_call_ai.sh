#!/bin/bash
# round 3, call ai: where the lane-per-channel linear demodulator overtakes the wavefront-per-channel one (bank size sweep)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for n in 0.003072 0.018432 0.073728 0.294912; do
  for w in 1 0; do
    CHZ_DEMOD_WAVE=$w timeout 300 python scripts/scale_kernels_probe.py $n linear 2>&1 | tail -1 | sed "s/^{/{\"CHZ_DEMOD_WAVE\": $w, /" | tee -a gpurun_out/r3_demod_crossover.jsonl
  done
done
