#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_dropin.py -m gpu -q --timeout 600 -x 2>&1 | tail -30 > gpurun_out/pytest_e.log
tail -30 gpurun_out/pytest_e.log
