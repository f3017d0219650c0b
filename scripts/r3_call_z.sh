#!/bin/bash
# round 3, call z: demodulator kernel with the host-side constants and the squared-up gain ramp -- parity, then cost per channel
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_golden.py -m gpu -q -x -k "demod or fm_ or coherent or linear or golden or filter2" -p no:cacheprovider 2>&1 | tail -3
for m in linear fm; do
  timeout 300 python scripts/scale_kernels_probe.py 1.5 $m 2>&1 | tail -1 | tee -a gpurun_out/r3_demod_consts.jsonl
done
