#!/bin/bash
# A/B of kernel variants selected by environment: KT_VARIANTS="A=1 B=2;A=3" (semicolon-separated env assignments)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/kt.log
IFS=';' read -ra VARS <<< "${KT_VARIANTS:-CHZ_TILES_PER_WG=1}"
for v in "${VARS[@]}"; do
  for rep in 1 2; do env $v timeout 120 python scripts/kernel_times.py "$v" 2>/dev/null | tail -1 >> gpurun_out/kt.log; done
done
cat gpurun_out/kt.log
