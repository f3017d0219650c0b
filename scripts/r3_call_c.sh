#!/bin/bash
# round 3, call C: the rest of the GPU tests, the drop-in legs and the wake-up knobs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3c_pytest.log
cat gpurun_out/r3c_pytest.log
: > gpurun_out/r3c_wake.txt
for args in "500 1024" "500 1024 KA9Q_HIP_FDOMAIN=0" "300 1024 KA9Q_HIP_WAKE=0,0 KA9Q_HIP_FDOMAIN=0" "300 1024 KA9Q_HIP_WAKE=8,4 KA9Q_HIP_FDOMAIN=0" "300 1024 KA9Q_HIP_WAKE=16,3 KA9Q_HIP_FDOMAIN=0" "300 1024 KA9Q_HIP_WAKE=32,2 KA9Q_HIP_FDOMAIN=0" "300 1024 KA9Q_HIP_WAKE=64,2 KA9Q_HIP_FDOMAIN=0" "300 1024 KA9Q_HIP_WAKE=1,2 KA9Q_HIP_FDOMAIN=0" "300 2000 KA9Q_HIP_FDOMAIN=0" "300 64 KA9Q_HIP_FDOMAIN=0"; do
  timeout 120 python scripts/dropin_rate.py $args 2>/dev/null | tail -1 >> gpurun_out/r3c_wake.txt
done
python - <<'PY'
import json
for ln in open("gpurun_out/r3c_wake.txt"):
    d = json.loads(ln)
    if "error" in d: print(d); continue
    print("%-45s thr %4d ms/block %.3f worst gap %.2f drops %d dev avg %.0f us" % (d["label"][:45], d["threads"], d["ms_per_block"], d["worst_block_gap_ms"], d["drops"], d["device_block_us_avg"]), d["front_end_us_per_block"], d["host_profile"])
PY
