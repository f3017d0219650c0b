#!/bin/bash
# first GPU call of a round: GPU tests, then the driver-style bench line and the default one
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err
timeout 600 python bench.py --no-crt --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -5 gpurun_out/pytest_gpu.log; tail -c 1500 gpurun_out/bench_driver.json; tail -c 600 gpurun_out/bench_driver.err
