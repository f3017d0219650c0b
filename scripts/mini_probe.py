#!/usr/bin/env python3
"""Cost of the pooled inline masters (radiod's filter2): one chz_mini_execute call = windows H2D, ONE launch (one workgroup
per instance), outputs D2H, for n instances of one geometry.  usage: mini_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load()
rng = np.random.default_rng(0)
out = {}
for L, M, n in ((240, 273, 1024), (960, 1089, 1024), (1920, 2177, 1024), (240, 273, 64)):
    N = L + M - 1
    pool = pkg.engine.MiniPool(L, M, n)
    insts = [pool.add() for _ in range(n)]
    for i in insts:
        pool.set_response(i, (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64) / N)
    win = (rng.standard_normal((n, N)) + 1j * rng.standard_normal((n, N))).astype(np.complex64)
    for _ in range(5):
        pool.execute(insts, win)
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        pool.execute(insts, win)
    dt = (time.perf_counter() - t0) / reps
    out["N=%d x %d" % (N, n)] = {"ms_per_call": dt * 1e3, "us_per_instance": dt * 1e6 / n, "host_bytes_per_call": n * (N + L) * 8}
    print("N=%d L=%d, %d instances: %.3f ms per call (%.2f us per instance; includes the ctypes marshalling of %d pointers)" % (N, L, n, dt * 1e3, dt * 1e6 / n, 2 * n))
    pool.close()
print(json.dumps(out))
