#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--no-crt --no-cpu-baseline"
timeout 300 python bench.py $B > gpurun_out/c_ticket.json 2> gpurun_out/c.err
BENCH_NO_NOTCH=1 timeout 300 python bench.py $B > gpurun_out/c_nonotch.json 2>> gpurun_out/c.err
CHZ_NOTCH_ORDER=unordered-timing-only timeout 300 python bench.py $B > gpurun_out/c_unordered.json 2>> gpurun_out/c.err
CHZ_STREAMS=2 timeout 300 python bench.py $B > gpurun_out/c_2streams.json 2>> gpurun_out/c.err
timeout 300 python bench.py $B --graph > gpurun_out/c_graph.json 2>> gpurun_out/c.err
BENCH_NO_NOTCH=1 timeout 300 python bench.py $B --graph > gpurun_out/c_graph_nonotch.json 2>> gpurun_out/c.err
for f in c_ticket c_nonotch c_unordered c_2streams c_graph c_graph_nonotch; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    j=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "ms/step %.4f min %.4f max %.4f reps %d gpu %.4f host %.4f" % (j["ms_per_step"],j["ms_per_step_min"],j["ms_per_step_max"],j["reps"],j["gpu_event_ms_per_step"],j["host_enqueue_ms_per_step"]), {k:round(v,2) for k,v in j["roofline"]["kernels_us"].items()})
except Exception as e: print(f,"ERR",e)
PY
done
tail -3 gpurun_out/c.err
