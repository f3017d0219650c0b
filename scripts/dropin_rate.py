#!/usr/bin/env python3
"""End-to-end rate THROUGH the filter.h drop-in (host ring -> H2D -> kernels -> D2H -> 1024 channel threads):
the PCIe-inclusive figure of DESIGN.md section 5.  usage: dropin_rate.py [nblocks]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import oracle_lib as ol
import test_dropin as td
nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
fs, L, M, olen = 129.6e6, 2592000, 648001, 240
N = L + M - 1
td._build_lib(); ol.build()
g = ol.SigGen(10.00002e6 / fs, 0.1, 0.01, ol.scale_ad(True, 1), True, seed=1)
x = np.tile(g.generate(8 * L), (nblocks + 7) // 8)[:nblocks * L]
kinds = [(50 / 12000, 3000 / 12000), (-200 / 12000, 200 / 12000), (-5000 / 12000, 5000 / 12000)]
plan = []
for i in range(nch):
    shift = ol.compute_tuning(N, fs, 1e6 + i * 60e3 + (i % 40))[1]
    lo, hi = kinds[i % 3]
    plan.append((shift, shift + 1, 10 ** 6, 10 ** 6, lo, hi, 11.0, lo, hi))
with tempfile.TemporaryDirectory() as tmp:
    env = {"HARNESS_RETUNE_MOD": os.environ["RETUNE_MOD"]} if os.environ.get("RETUNE_MOD") else None      # 1/RETUNE_MOD of the channels retune every block
    out, spec, meta = td._run_harness(tmp, L, M, ol.REAL, olen, plan, nblocks, int(os.environ.get("CHUNK", "16384")), x, env=env)
el = float(meta["elapsed_s"])
print("drop-in, config 3 through filter.h (%d pthreads, retune 1/%s per block)" % (nch, os.environ.get("RETUNE_MOD", "inf")) + ": %d blocks in %.3f s = %.2f ms/block = %.1fx real time; "
      "device block time avg %.1f us max %.1f us; drops %s" % (nblocks, el, el / nblocks * 1e3, 0.02 / (el / nblocks),
      int(meta["avg_block_ns"]) / 1e3, int(meta["max_block_ns"]) / 1e3, meta["drops"]))
