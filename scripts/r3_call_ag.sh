#!/bin/bash
# round 3, call ag: the demodulator kernels' double-precision wave reductions through DPP -- parity, then FM's cost per channel
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_golden.py -m gpu -q -x -k "demod or coherent or linear or golden or filter2 or partial_rerun or fm_" -p no:cacheprovider 2>&1 | tail -2
timeout 300 python scripts/scale_kernels_probe.py 1.5 fm 2>&1 | tail -1 | tee -a gpurun_out/r3_fm_probe_lanes3.jsonl
timeout 300 python scripts/scale_kernels_probe.py 1.5 fmtone 2>&1 | tail -1 | tee -a gpurun_out/r3_fm_probe_lanes3.jsonl
