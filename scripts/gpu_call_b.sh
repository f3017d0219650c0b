#!/bin/bash
# notch ordering A/B + profiler run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--no-crt --no-cpu-baseline"
timeout 300 python bench.py $B > gpurun_out/b_ticket.json 2> gpurun_out/b_ticket.err
CHZ_NOTCH_ORDER=event timeout 300 python bench.py $B > gpurun_out/b_event.json 2> gpurun_out/b_event.err
timeout 300 python bench.py $B --steps 20 --warmup 5 > gpurun_out/b_ticket20.json 2>> gpurun_out/b_ticket.err
CHZ_ENQ_THREADS=1 timeout 300 python bench.py $B > gpurun_out/b_ticket_1thr.json 2>> gpurun_out/b_ticket.err
CHZ_ENQ_THREADS=4 timeout 300 python bench.py $B > gpurun_out/b_ticket_4thr.json 2>> gpurun_out/b_ticket.err
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "notch or retune or rccl or swaps or dropin or graph" 2>&1 | tail -15 > gpurun_out/pytest_b.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_ticket -o t -- python $GRAFT_REPO_ROOT/bench.py $B --steps 100 --warmup 20 --min-seconds 0.1 > $GRAFT_REPO_ROOT/gpurun_out/b_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/b_prof.err
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py gpurun_out/prof_ticket > gpurun_out/prof_ticket_summary.txt 2>&1
for f in b_ticket b_event b_ticket20 b_ticket_1thr b_ticket_4thr b_prof; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    j=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "ms/step %.4f min %.4f max %.4f reps %d gpu %.4f host %.4f" % (j["ms_per_step"],j["ms_per_step_min"],j["ms_per_step_max"],j["reps"],j["gpu_event_ms_per_step"],j["host_enqueue_ms_per_step"]), {k:round(v,2) for k,v in j["roofline"]["kernels_us"].items()})
except Exception as e: print(f,"ERR",e)
PY
done
tail -3 gpurun_out/pytest_b.log; head -30 gpurun_out/prof_ticket_summary.txt
