#!/bin/bash
# round 3, call ah: bench contract test with the next_rows leg, then the driver's command
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_bench_contract.py -m gpu -q -x -k "one_json_line" -p no:cacheprovider 2>&1 | tail -3
( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
tail -4 gpurun_out/r03_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d['next_rows'], indent=1))
"
