#!/bin/bash
# round 3, call aj: demodulator tests on both paths on the device; bench contract; chain leg
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_pipeline.py tests/test_golden.py tests/test_bench_contract.py -m gpu -q -x -k "demod or coherent or linear or golden or filter2 or partial_rerun or fm_ or one_json_line" -p no:cacheprovider 2>&1 | tail -3
