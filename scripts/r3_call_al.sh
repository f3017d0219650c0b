#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 300 python scripts/scale_kernels_probe.py 1.5 fmvar 2>&1 | tail -1 | tee -a gpurun_out/r3_fm_probe_lanes3.jsonl
