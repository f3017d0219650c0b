#!/usr/bin/env python3
"""EXPERIMENT check (round 5): the batched forward passes (libchz_hip_batch.so, CHZ_FWD_BATCH_N=2|4) give the spectra the shipped
one-block-per-launch passes give -- 8 blocks of noise through chz_run_blocks with idle banks, every slot of the last four blocks
against the float64 oracle.   usage: CHZ_LIB=.../libchz_hip_batch.so CHZ_FWD_BATCH_N=4 python scripts/batch_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from conftest import load_pkg

pkg = load_pkg()
L, M = 2592000, 648001
rng = np.random.default_rng(3)
eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
x = rng.standard_normal(8 * L).astype(np.float32)
eng.write(x[:8 * L - (M - 1)]); eng.write(x[8 * L - (M - 1):])
eng.run_blocks(0, 8)
st = ol.Stream(L, M, ol.REAL)
st.push(x[7 * L:], f64=True)                 # the ring was filled to the brim: block 0's history is the end of the ring
worst = 0.0
for job in range(8):
    want = st.push(x[job * L:(job + 1) * L], f64=True)
    if job >= 4:
        got = eng.spectrum(job % 4)
        worst = max(worst, float(np.linalg.norm(got - want) / np.linalg.norm(want)))
eng.close()
print("batch_check CHZ_FWD_BATCH_N=%s worst rel-L2 of blocks 4..7 = %.3g  %s" % (os.environ.get("CHZ_FWD_BATCH_N", "-"), worst, "OK" if worst < 1e-6 else "MISMATCH"))
sys.exit(0 if worst < 1e-6 else 1)
