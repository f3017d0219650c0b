#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python scripts/mini_probe.py 2>/dev/null | grep -v "^{" > gpurun_out/i_mini.txt
timeout 300 python bench.py --config 2 --no-crt --no-cpu-baseline > gpurun_out/i_cfg2.json 2>/dev/null
BENCH_FORCE_DIST=1 MASTER_PORT=29655 timeout 300 python bench.py --config 4 --no-crt --no-cpu-baseline > gpurun_out/i_cfg4.json 2>/dev/null
timeout 300 python bench.py --config 4 --no-cpu-baseline --crt-blocks 500 > gpurun_out/i_cfg4_crt.json 2>/dev/null
timeout 300 python bench.py --config 5 --no-crt --no-cpu-baseline > gpurun_out/i_cfg5.json 2>/dev/null
cat gpurun_out/i_mini.txt
for f in i_cfg2 i_cfg4 i_cfg4_crt i_cfg5; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    j=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "value %.0f ms/step %.5f GBps %.0f exch %s legs %s" % (j["value"], j["ms_per_step"], j["step_algorithmic_GBps"], (j["exchange"] or "")[:60], {k: round(v["ms_per_step"],5) for k,v in (j["legs"] or {}).items()}))
    if j.get("c_rt"): print("   c_rt", {k:j["c_rt"].get(k) for k in ("channels","P","blocks","worst_block_ms","mean_block_ms","sustained","dram_side_GBps")})
except Exception as e: print(f,"ERR",e)
PY
done
