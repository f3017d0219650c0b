#!/bin/bash
# round 3, call B: GPU tests, the full bench line with its new legs, and the drop-in's wake-up knobs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3b_pytest.log
cat gpurun_out/r3b_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r3b_bench.err | tail -1 > gpurun_out/r3b_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3b_bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print("c_rt", json.dumps(d["c_rt"])[:900])
print("dropin", json.dumps(d["dropin"])[:3000])
print("c_rt_pcie", json.dumps(d["c_rt_pcie"])[:1500])
print("cpu", json.dumps(d["cpu_baseline"])[:1200])
PY
: > gpurun_out/r3b_wake.txt
for w in "0,2" "0,0" "8,4" "16,3" "32,2" "64,2" "1,2"; do
  timeout 120 python scripts/dropin_rate.py 300 1024 KA9Q_HIP_WAKE=$w KA9Q_HIP_FDOMAIN=0 2>/dev/null | tail -1 >> gpurun_out/r3b_wake.txt
done
python - <<'PY'
import json
for ln in open("gpurun_out/r3b_wake.txt"):
    d = json.loads(ln)
    print(d["label"], "ms/block %.3f worst gap %.2f" % (d["ms_per_block"], d["worst_block_gap_ms"]), d["front_end_us_per_block"], d["host_profile"])
PY
nproc
