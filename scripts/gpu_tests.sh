#!/bin/bash
# the driver's GPU tier: all GPU tests, then smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 ${PYTEST_ARGS:-} 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
tail -25 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log
