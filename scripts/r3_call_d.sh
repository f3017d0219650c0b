#!/bin/bash
# round 3, call D: how many wake words for the drop-in's channel threads
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/r3d_wake.txt
for sh in 1 4 8 16 32 64; do for thr in 1024 2000; do
  timeout 120 python scripts/dropin_rate.py 500 $thr KA9Q_HIP_FDOMAIN=0 KA9Q_HIP_WAKE_SHARDS=$sh 2>/dev/null | tail -1 >> gpurun_out/r3d_wake.txt
done; done
timeout 120 python scripts/dropin_rate.py 500 1024 KA9Q_HIP_FDOMAIN=0 KA9Q_HIP_WAKE=0,0 2>/dev/null | tail -1 >> gpurun_out/r3d_wake.txt
timeout 120 python scripts/dropin_rate.py 500 1024 2>/dev/null | tail -1 >> gpurun_out/r3d_wake.txt
timeout 120 python scripts/dropin_rate.py 500 64 2>/dev/null | tail -1 >> gpurun_out/r3d_wake.txt
python - <<'PY'
import json
for ln in open("gpurun_out/r3d_wake.txt"):
    d = json.loads(ln)
    if "error" in d: print(d); continue
    print("%-50s thr %4d ms/block %.3f worst gap %.2f drops %d dev avg %.0f us" % (d["label"][:50], d["threads"], d["ms_per_block"], d["worst_block_gap_ms"], d["drops"], d["device_block_us_avg"]), d["front_end_us_per_block"], "cb->slave mean %.0f worst %.0f" % (d["host_profile"]["callback_to_slave_has_its_block_us_mean"], d["host_profile"]["callback_to_slave_has_its_block_us_worst"]))
PY
