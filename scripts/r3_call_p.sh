#!/bin/bash
# round 3, call P: whole GPU suite, the full bench line, rocprofv3 kernel traces and PMC passes of the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed|error" | tail -6 | tee gpurun_out/r03_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r03_bench.err | tail -1 > gpurun_out/r03_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "host_enqueue", d["host_enqueue_ms_per_step"])
print("roofline frac", d["roofline"]["frac"], "pipelined", d["roofline"]["pipelined"]["frac"], "kernels", d["roofline"]["kernels_us"])
print("c_rt", d["c_rt"]["channels"], d["c_rt"]["worst_block_ms"], [ (p["channels"], round(p["worst_block_ms"],2), p["sustained"]) for p in d["c_rt"]["probes"]])
for x in d["dropin"]: print("dropin", x.get("threads"), x.get("ms_per_block"), x.get("worst_block_gap_ms"), x.get("drops"), x.get("error"))
for x in d["c_rt_pcie"]: print("pcie", x.get("channels"), x.get("worst_block_ms"), x.get("mean_block_ms"), x.get("d2h_GBps"), x.get("error"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("fft_calibration"))
PY
timeout 1500 bash scripts/gpu_profile.sh r03 2>&1 | tail -25
