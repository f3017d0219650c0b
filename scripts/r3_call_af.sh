#!/bin/bash
# round 3, call af: C_rt ladder above 20.5 M
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 --crt-ladder 20.5,21.0,21.5 --no-dropin --no-crt-pcie --no-cpu-baseline > gpurun_out/r3_crt_top.json 2> gpurun_out/r3_crt_top.err
tail -2 gpurun_out/r3_crt_top.err
python -c "
import json
d=json.loads(open('gpurun_out/r3_crt_top.json').read().strip().splitlines()[-1])
c=d['c_rt']; print(c.get('error')); print(c.get('channels'), [(p['channels'], round(p['worst_block_ms'],2), round(p['mean_block_ms'],2), p['sustained']) for p in c.get('probes',[])])
"
