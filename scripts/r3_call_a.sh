#!/bin/bash
# round 3, call A: GPU tests of the tree + eager vs hipGraph (ticketed notches inside the capture)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3a_pytest.log
cat gpurun_out/r3a_pytest.log
timeout 200 python bench.py --steps 20 --warmup 5 --no-crt --no-cpu-baseline 2>gpurun_out/r3a_eager.err | tail -1 > gpurun_out/r3a_eager.json
for gb in 8 32 128; do
  CHZ_GRAPH_BLOCKS=$gb timeout 200 python bench.py --steps 20 --warmup 5 --no-crt --no-cpu-baseline --graph 2>gpurun_out/r3a_graph$gb.err | tail -1 > gpurun_out/r3a_graph$gb.json
done
CHZ_GRAPH_NOTCH=event timeout 200 python bench.py --steps 20 --warmup 5 --no-crt --no-cpu-baseline --graph 2>/dev/null | tail -1 > gpurun_out/r3a_graph_event.json
python - <<'PY'
import json
for n in ("eager", "graph8", "graph32", "graph128", "graph_event"):
    try:
        d = json.load(open("gpurun_out/r3a_%s.json" % n))
        print(n, "ms_per_step %.5f  host_enqueue %.5f  gpu_event %.5f  pipelined_fwd_us %.2f" % (d["ms_per_step"], d["host_enqueue_ms_per_step"], d["gpu_event_ms_per_step"], d["roofline"]["pipelined"]["forward_us_per_block"]))
    except Exception as ex:
        print(n, "FAILED", ex)
PY
