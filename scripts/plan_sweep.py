#!/usr/bin/env python3
"""Time the forward transform for a list of plan specs on the GPU (forward only, no channel banks).
usage: plan_sweep.py [N_L] spec1 spec2 ...   (L defaults to 2592000 -> N = 3,240,000)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load()
args = sys.argv[1:]
L = 2592000
if args and args[0].isdigit():
    L = int(args.pop(0))
M = L // 4 + 1
x = np.random.default_rng(0).standard_normal(8 * L).astype(np.float32)
for spec in args:
    try:
        eng = pkg.engine.Engine(L, M, pkg.engine.REAL, plan=spec, ring_blocks=8)
    except Exception as e:
        print("%-28s FAILED %s" % (spec, e)); continue
    eng.write(x[:8 * L - (M - 1)]); eng.write(x[8 * L - (M - 1):])
    eng.run_blocks(0, 64, graph=False)
    it = eng.run_blocks(0, 200, graph=False, instrument=True)
    t = eng.run_blocks(0, 800, graph=False)
    k = [it.first_ms / max(it.first_n, 1) * 1e3, it.cols_ms / max(it.cols_n, 1) * 1e3, it.rows_ms / max(it.rows_n, 1) * 1e3]
    print("%-28s total %6.2f us/blk (lanes)  kernels(ev) first %5.1f cols %5.1f rows %5.1f sum %5.1f | %s" %
          (spec, t.total_ms / 800 * 1e3, k[0], k[1], k[2], sum(k), eng.plan.split("tiles")[1]))
    eng.close()
