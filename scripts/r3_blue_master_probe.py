#!/usr/bin/env python3
"""A master whose length is outside the compiled axes (chirp-z): plan, accuracy against numpy's float64 FFT, time per block."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load()
for L, M in ((2592000, 648002), (2592000, 648001), (48000, 12002)):
    N = L + M - 1
    eng = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=8)
    x = np.random.default_rng(1).standard_normal(8 * L).astype(np.float32)
    eng.write(x[:8 * L - (M - 1)]); eng.write(x[8 * L - (M - 1):])
    eng.run_blocks(0, 8)
    t = eng.run_blocks(8, 32)
    got = eng.spectrum((8 + 31) % 4)
    j = 39 % 8
    win = np.concatenate([x[((j * L - (M - 1)) % (8 * L)):], x])[:N] if j * L - (M - 1) < 0 else x[j * L - (M - 1):j * L - (M - 1) + N]
    want = np.fft.rfft(win.astype(np.float64))
    err = float(np.sqrt(np.sum(np.abs(got - want) ** 2) / np.sum(np.abs(want) ** 2)))
    print(json.dumps({"L": L, "M": M, "N": N, "plan": eng.plan, "us_per_block": t.total_ms / 32 * 1e3, "spectrum_rel_l2_vs_f64": err}))
    eng.close()
