#!/bin/bash
# round 3, call am: FM at one channel per lane -- parity on the device (both paths), cost per channel
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_golden.py -m gpu -q -x -k "demod or coherent or linear or golden or filter2 or partial_rerun or fm_" -p no:cacheprovider 2>&1 | tail -3
for m in fm fmvar fmtone; do
  timeout 300 python scripts/scale_kernels_probe.py 1.5 $m 2>&1 | tail -1 | sed "s/^{/{\"path\": \"demod_fm_lanes\", /" | tee -a gpurun_out/r3_fm_lanes.jsonl
done
