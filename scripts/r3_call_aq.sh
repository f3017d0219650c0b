#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
PROBE_SYNC=1 timeout 500 python scripts/r3_pcie_pipelined_probe.py 1216512 1320960 1428480 2>&1 | tail -3 | tee gpurun_out/r3_pcie_sync.jsonl
