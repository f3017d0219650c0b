#!/bin/bash
# round 3, call s: noise_est with the exponent found by bisection over the window's own range, DPP reductions, two-register mantissa path
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -q -x -k "noise or demod or fm_ or linear" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python scripts/scale_kernels_probe.py 1.5 linear 2>&1 | tail -1 | tee -a gpurun_out/r3_noise_energy.jsonl
