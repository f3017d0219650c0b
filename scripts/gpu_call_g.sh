#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/next_rows_probe.py 2>/dev/null | grep "^tuned" > gpurun_out/g_next_rows.txt
CRT_BLOCKS=100 timeout 900 python scripts/crt_pcie_probe.py 0.8 2>/dev/null | tail -1 > gpurun_out/g_crt_pcie.json
CRT_BLOCKS=100 CRT_DEMOD=1 timeout 900 python scripts/crt_pcie_probe.py 2.4 2>/dev/null | tail -1 > gpurun_out/g_crt_pcie_demod.json
cat gpurun_out/g_next_rows.txt; cut -c1-600 gpurun_out/g_crt_pcie.json; echo; cut -c1-900 gpurun_out/g_crt_pcie_demod.json
