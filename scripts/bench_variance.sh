#!/bin/bash
# bench.py repeated with 1/2/4 issuing host threads: block time, host enqueue wall time, GPU event time (us)
for t in 2 1 2 4 2 1; do
  CHZ_ENQ_THREADS=$t python bench.py --no-cpu-baseline --no-crt 2>/dev/null | tail -1 | T=$t python -c "import os,sys,json; j=json.loads(sys.stdin.read()); print('threads', os.environ['T'], round(j['ms_per_step']*1e3,2), round(j['host_enqueue_ms_per_step']*1e3,2), round(j['gpu_event_ms_per_step']*1e3,2))"
done
