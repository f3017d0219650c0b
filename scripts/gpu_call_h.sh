#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x --timeout 300 -k "demod" 2>&1 | tail -5 > gpurun_out/h_pytest.log
timeout 600 python scripts/next_rows_probe.py 2>/dev/null | grep "^tuned" > gpurun_out/g_next_rows.txt
CRT_BLOCKS=100 CRT_DEMOD=1 timeout 900 python scripts/crt_pcie_probe.py 3.6 2>/dev/null | tail -1 > gpurun_out/g_crt_pcie_demod.json
tail -3 gpurun_out/h_pytest.log; cat gpurun_out/g_next_rows.txt | cut -c1-260; cut -c1-1100 gpurun_out/g_crt_pcie_demod.json
