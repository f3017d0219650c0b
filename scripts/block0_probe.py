#!/usr/bin/env python3
"""Round 6: where does block 0 go for the FIRST drop-in process on a fresh box?  (driver's round-5 record: 14.5 ms at 1024 threads in the
first leg, 5.8 ms at 2000 threads right after.)  Runs bench.py's paced drop-in leg -- the C harness, a process of its own -- as the first
GPU user of this box, then again, then with 2000 threads, with the drop-in's per-block profile of blocks 0..7 (KA9Q_HIP_PROFILE=1:
enqueue -> callback on the device, time inside execute_filter_input, slowest slave, staged hits / misses).  No torch, no GPU use in
this process.   usage: python scripts/block0_probe.py [blocks] > gpurun_out/block0_probe.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import oracle_lib as ol

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ol.build()
wl = bench.workload_for(4, 0, 1, 1024) if os.environ.get("BLOCK0_CONFIG4") else bench.workload_for(3, 0, 1, 1024)
ring = bench.siggen_ring(ol, wl["fs"], l=wl["L"], real=wl.get("real", True))
out = []
only = os.environ.get("BLOCK0_ONLY", "")        # "1": only the BLOCK0_EXTRA configurations; "2": one default 1024-thread run
base = [] if only == "1" else [("first process on the box, 1024 threads", 1024, {}), ("second process, 1024 threads", 1024, {}),
                      ("third process, 2000 threads, noise from the device", 2000, {"KA9Q_HIP_FDOMAIN": "0", "KA9Q_HIP_NOISE_SAMPRATE": "%.1f" % wl["fs"]}),
                      ("fourth process, 1024 threads", 1024, {})]
if only == "2":
    base = base[:1]
for label, n, env in base + [(l, n, dict(e)) for l, n, e in json.loads(os.environ.get("BLOCK0_EXTRA", "[]"))]:
    r = bench.dropin_leg(wl, ring, n, blocks, env, label, paced_us=20000)
    p = r.get("paced") or {}
    rec = {"label": label, "threads": n, "drops": r.get("drops"), "block0_ms": p.get("block0_ms"), "first_8_ms": r.get("first_8_blocks_latency_ms"),
           "p50_ms": p.get("p50_ms"), "p99_ms": p.get("p99_ms"), "max_ms": p.get("max_ms"), "first_blocks_profile": r.get("first_blocks_profile"),
           "process_wall_s": r.get("process_wall_s"), "error": r.get("error")}
    out.append(rec)
    print(json.dumps(rec), flush=True)
