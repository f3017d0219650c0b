#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
CRT_BLOCKS=500 timeout 1500 python scripts/crt_probe.py 20.5 2>/dev/null | tail -1 > gpurun_out/j_crt.json
cut -c1-1500 gpurun_out/j_crt.json
