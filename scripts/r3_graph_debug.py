#!/usr/bin/env python3
"""Does a hipGraph replay continue the notch recurrence?  Eager engine = the reference (its tests pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load()
L, M = 25920, 6481
rng = np.random.default_rng(52)
ring = (rng.standard_normal(8 * L) + 0.3).astype(np.float32)
bins = [0, 125, 16000]

def run(plan, env=None):
    for k, v in (env or {}).items(): os.environ[k] = v
    eng = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=8)
    for k in (env or {}): os.environ.pop(k)
    eng.set_notches(bins, 0.05)
    eng.write(ring[:8 * L - (M - 1)]); eng.write(ring[8 * L - (M - 1):])
    job = 0
    for graph, n in plan:
        eng.run_blocks(job, n, graph=graph); job += n
    got = eng.spectrum((job - 1) % 4)
    eng.close()
    return got

def cmp(name, plan, env=None):
    want = run([(False, sum(n for _, n in plan))])
    got = run(plan, env)
    d = np.abs(got - want)
    print("%-40s notch bins %s  rest max %.3g  (|want| at bins %s)" % (name, ["%.3g" % d[b] for b in bins], np.delete(d, bins).max(), ["%.3g" % abs(want[b]) for b in bins]))


full = [(True, 40), (False, 13), (True, 19), (True, 64), (False, 3)]
cmp("test plan", full)
cmp("test plan as eager pieces", [(False, n) for _, n in full])
for k in range(1, len(full)):
    cmp("test plan, first %d" % k, full[:k])
cmp("eager 53 + graph 19", [(False, 53), (True, 19)])
cmp("eager 5 + graph 16", [(False, 5), (True, 16)])
cmp("eager 5 + graph 8", [(False, 5), (True, 8)])
cmp("eager 4 + graph 8", [(False, 4), (True, 8)])
cmp("eager 1 + graph 8", [(False, 1), (True, 8)])
