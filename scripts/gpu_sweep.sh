#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python scripts/plan_sweep.py $SWEEP_PLANS 2>/dev/null > gpurun_out/sweep.txt
sort -k3 -n gpurun_out/sweep.txt | head -40
