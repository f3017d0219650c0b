#!/bin/bash
# round 3, call G: twiddle A/B, two-thread sharded loop (single-rank RCCL), whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/r3g_tw.txt
for v in 0 1 0 1; do CHZ_TW_SHUFFLE=$v timeout 120 python scripts/kernel_times.py "CHZ_TW_SHUFFLE=$v" 2>/dev/null | tail -1 >> gpurun_out/r3g_tw.txt; done
cat gpurun_out/r3g_tw.txt
for t in 1 2; do CHZ_ENQ_THREADS=$t BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --config 4 --exchange broadcast --no-crt --no-cpu-baseline --no-dropin --no-crt-pcie 2>/dev/null | tail -1 > gpurun_out/r3g_dist_thr$t.json; done
python - <<'PY'
import json
for t in (1, 2):
    d = json.load(open("gpurun_out/r3g_dist_thr%d.json" % t))
    print("issuing threads", t, "ms_per_step %.5f host_enqueue %.5f" % (d["ms_per_step"], d["host_enqueue_ms_per_step"]), "rccl_ranks", d.get("rccl_ranks"), {k: round(v["ms_per_step"], 5) for k, v in (d.get("legs") or {}).items()})
PY
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
