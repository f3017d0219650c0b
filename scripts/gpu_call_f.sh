#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dropin.py -m gpu -q -x --timeout 300 2>&1 | tail -5 > gpurun_out/f_pytest.log
: > gpurun_out/f_dropin.log
timeout 300 python scripts/dropin_rate.py 60 1024 2>/dev/null | tail -1 >> gpurun_out/f_dropin.log
RETUNE_MOD=20 timeout 300 python scripts/dropin_rate.py 60 1024 2>/dev/null | tail -1 >> gpurun_out/f_dropin.log
CHUNK=2592000 timeout 300 python scripts/dropin_rate.py 60 1024 2>/dev/null | tail -1 >> gpurun_out/f_dropin.log
CHUNK=2592000 RETUNE_MOD=20 timeout 300 python scripts/dropin_rate.py 60 1024 2>/dev/null | tail -1 >> gpurun_out/f_dropin.log
CHUNK=2592000 RETUNE_MOD=20 KA9Q_HIP_FDOMAIN=0 timeout 300 python scripts/dropin_rate.py 60 1024 2>/dev/null | tail -1 >> gpurun_out/f_dropin.log
timeout 300 python scripts/dropin_rate.py 60 64 2>/dev/null | tail -1 >> gpurun_out/f_dropin.log
tail -3 gpurun_out/f_pytest.log; cat gpurun_out/f_dropin.log
