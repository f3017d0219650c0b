#!/bin/bash
# round 3, call u: FM's PLL demodulator and PL-tone detector one channel per lane -- parity (incl. bit identity with the one-kernel path), then cost per channel
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_golden.py -m gpu -q -x -k "demod or fm_ or coherent or linear or golden or filter2" -p no:cacheprovider 2>&1 | tail -5
for m in fm fmtone fmpll; do
  timeout 300 python scripts/scale_kernels_probe.py 0.3 $m 2>&1 | tail -1 | tee -a gpurun_out/r3_fm_probe_lanes.jsonl
done
timeout 300 python scripts/scale_kernels_probe.py 1.0 fmpll 2>&1 | tail -1 | tee -a gpurun_out/r3_fm_probe_lanes.jsonl
timeout 300 python scripts/scale_kernels_probe.py 1.5 fmtone 2>&1 | tail -1 | tee -a gpurun_out/r3_fm_probe_lanes.jsonl
