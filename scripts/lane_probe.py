#!/usr/bin/env python3
"""How well do blocks overlap across HIP streams?  Forward + channels, with and without the notch
(the notch state serialises fwd_rows of consecutive blocks), eager vs hipGraph, host enqueue time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load()
L = 2592000; M = L // 4 + 1
x = np.random.default_rng(0).standard_normal(8 * L).astype(np.float32)
for notch in (True, False):
    for graph in (False, True):
        eng = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=8)
        eng.write(x[:8 * L - (M - 1)]); eng.write(x[8 * L - (M - 1):])
        b = eng.bank(300, 240, 1024)
        b.set_responses(0, np.ones((1024, 300), np.complex64)); b.set_shifts(0, 25000 + np.arange(1024) * 1500); b.set_active(1024)
        if notch: eng.set_notches([0], 0.01)
        eng.run_blocks(0, 160, graph=graph)
        t = eng.run_blocks(160, 1600, graph=graph)
        print("lanes=%d notch=%d graph=%d  %.2f us/block  host enqueue %.2f us/block" % (eng.lanes, notch, graph, t.total_ms / 1600 * 1e3, t.enqueue_ms / 1600 * 1e3))
        eng.close()
