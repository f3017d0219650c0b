#!/usr/bin/env python3
"""End-to-end rate THROUGH the filter.h drop-in (host ring -> H2D -> kernels -> D2H -> channel pthreads): bench.py's `dropin`
leg on its own, for A/B runs of the host-side knobs.
usage: dropin_rate.py [nblocks] [nthreads] [KEY=VALUE ...]      e.g.  dropin_rate.py 500 1024 KA9Q_HIP_WAKE=16,3 KA9Q_HIP_FDOMAIN=0"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import oracle_lib
args = [a for a in sys.argv[1:] if "=" not in a]
env = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
nblocks = int(args[0]) if len(args) > 0 else 500
nthreads = int(args[1]) if len(args) > 1 else 1024
wl = bench.workload_for(3, 0, 1, 0)
ring = bench.siggen_ring(oracle_lib, wl["fs"], wl["seed"], wl["L"])
r = bench.dropin_leg(wl, ring, nthreads, nblocks, env, " ".join("%s=%s" % kv for kv in env.items()) or "defaults")
print(json.dumps(r))
