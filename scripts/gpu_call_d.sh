#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--no-crt --no-cpu-baseline"
(cd old_r1 && timeout 300 python bench.py $B --steps 2000 --warmup 200 > ../gpurun_out/d_old2000.json 2> ../gpurun_out/d.err)
(cd old_r1 && timeout 300 python bench.py $B --steps 200 --warmup 50 > ../gpurun_out/d_old200.json 2>> ../gpurun_out/d.err)
timeout 300 python bench.py $B --steps 2000 --warmup 200 > gpurun_out/d_new2000.json 2>> gpurun_out/d.err
BENCH_NO_NOTCH=1 timeout 300 python bench.py $B --steps 2000 --warmup 200 > gpurun_out/d_new2000_nonotch.json 2>> gpurun_out/d.err
CHZ_NOTCH_ORDER=event timeout 300 python bench.py $B --steps 2000 --warmup 200 > gpurun_out/d_new2000_event.json 2>> gpurun_out/d.err
timeout 300 python bench.py $B > gpurun_out/d_new200.json 2>> gpurun_out/d.err
timeout 300 python bench.py $B --steps 20 --warmup 5 > gpurun_out/d_new20.json 2>> gpurun_out/d.err
(cd old_r1 && timeout 300 python bench.py $B --steps 2000 --warmup 200 > ../gpurun_out/d_old2000b.json 2>> ../gpurun_out/d.err)
for f in d_old2000 d_old200 d_new2000 d_new2000_nonotch d_new2000_event d_new200 d_new20 d_old2000b; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    j=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "ms/step %.4f gpu %.4f host %.4f" % (j["ms_per_step"],j["gpu_event_ms_per_step"],j["host_enqueue_ms_per_step"]), {k:round(v,2) for k,v in j["roofline"]["kernels_us"].items()})
except Exception as e: print(f,"ERR",e)
PY
done
tail -3 gpurun_out/d.err
