#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 500 python scripts/r3_pcie_pipelined_probe.py 1430000 1540000 1650000 2>&1 | tail -3 | tee gpurun_out/r3_pcie_pipelined.jsonl
